// bf16 attention on v_mfma_f32_32x32x16_bf16 (gfx950).  Same structure as attn.hip (scores transposed,
// S^T = K Q^T, whole key range in accumulators, softmax lane-local), with bf16 operands end to end:
//   * Q, K chunks and V chunks are staged to LDS as bf16 rows ([row][hd], 16-byte row pad);
//   * S^T tiles: A = K rows (ds_read_b128 fragments), B = Q rows (kept in registers for all chunks);
//   * P = softmax(S) stays in the accumulator registers, is normalised, converted to bf16 in place and
//     becomes the A operand of O = P V (accumulator rows = k index; k order inside a 16-step is
//     16s + 8(e>>2) + 4h + (e&3));
//   * the matching B operand V[key][dim] needs, per lane (fixed dim), 4 consecutive keys: exactly what
//     ds_read_b64_tr_b16 delivers from the row-major V image -- two transposed reads per fragment,
//     no transposing stores.
#include <cstdio>

#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int HDT, int NCH>
__global__ __launch_bounds__(256) void attn_bf16_kernel(AttnP p) {
    constexpr int HD = HDT * 32;
    constexpr int ROWB = HD * 2 + 16;  // LDS row stride in bytes
    constexpr int NS = HD / 16;        // MFMA k-steps over the head dim
    constexpr int CPR = HD / 8;        // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int b = blockIdx.x, head = blockIdx.y, qg = blockIdx.z;
    const int q0 = qg * 128;
    const int Lk = p.L1 + p.L2;
    const int l31 = lane & 31, lh = lane >> 5;

    const bf16_t* Qb = (const bf16_t*)p.Q + b * p.q_bstride + head * HD;
    const bf16_t* K1 = (const bf16_t*)p.K1 + b * p.kv1_bstride + head * HD;
    const bf16_t* V1 = (const bf16_t*)p.V1 + b * p.kv1_bstride + head * HD;
    const bf16_t* K2 = p.K2 ? (const bf16_t*)p.K2 + head * HD : nullptr;
    const bf16_t* V2 = p.V2 ? (const bf16_t*)p.V2 + head * HD : nullptr;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // ---- Q: stage nw*32 rows, keep this wave's B fragments in registers
    // query slots: [0, Lq) from Q (per batch element), [Lq1p, Lq1p + Lq2) from the shared segment Q2
    const int Lq1p = (p.Lq + 31) & ~31;
    const bf16_t* Q2b = p.Q2 ? (const bf16_t*)p.Q2 + head * HD : nullptr;
    for (int c = tid; c < nw * 32 * CPR; c += nthr) {
        const int r = c / CPR, kc = c % CPR, qi = q0 + r;
        u32x4 v = zero4;
        if (qi < p.Lq)
            v = *(const u32x4*)(Qb + (long long)qi * p.ldq + kc * 8);
        else if (Q2b && qi >= Lq1p && qi - Lq1p < p.Lq2)
            v = *(const u32x4*)(Q2b + (long long)(qi - Lq1p) * p.ldq2 + kc * 8);
        *(u32x4*)(lds + r * ROWB + kc * 16) = v;
    }
    __syncthreads();
    u32x4 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *(const u32x4*)(lds + (wid * 32 + l31) * ROWB + 32 * s + 16 * lh);
    __syncthreads();

    f32x16 sacc[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[c][jt][e] = 0.f;

    // ---- S^T = K Q^T, 64 keys per staged chunk
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j0 = c * 64;
        if (j0 < Lk) {
            for (int x = tid; x < 64 * CPR; x += nthr) {
                const int r = x / CPR, kc = x % CPR, j = j0 + r;
                u32x4 v = zero4;
                if (j < p.L1)
                    v = *(const u32x4*)(K1 + (long long)j * p.ldkv1 + kc * 8);
                else if (j < Lk)
                    v = *(const u32x4*)(K2 + (long long)(j - p.L1) * p.ldkv2 + kc * 8);
                *(u32x4*)(lds + r * ROWB + kc * 16) = v;
            }
            __syncthreads();
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                if (j0 + jt * 32 < Lk) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const u32x4 kf = *(const u32x4*)(lds + (jt * 32 + l31) * ROWB + 32 * s + 16 * lh);
                        sacc[c][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), sacc[c][jt], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- softmax over keys (register index + lane half) for the lane's query
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = c * 64 + jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = (j < Lk) ? sacc[c][jt][e] * p.scale : -INFINITY;
                sacc[c][jt][e] = v;
                m = fmaxf(m, v);
            }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = __builtin_amdgcn_exp2f((sacc[c][jt][e] - m) * 1.44269504088896340736f);
                sacc[c][jt][e] = v;
                l += v;
            }
    l += __shfl_xor(l, 32);
    float inv = 1.0f / l;
    float* fbuf = (float*)(lds + 64 * ROWB);  // per-query weight of the pre-reduced key block (behind the K/V image)
    if (p.pre_m) {
        // merge with the pre-reduced (batch-independent) key block: m_t = max(m, m_pre),
        // l_t = l e^{m - m_t} + l_pre e^{m_pre - m_t};  P gets e^{m - m_t} / l_t, the block's pre_O gets e^{m_pre - m_t} / l_t
        const int qi = q0 + wid * 32 + l31;
        float mp = -INFINITY, lp = 0.f;
        if (qi < p.Lq) {
            mp = p.pre_m[head * p.Lq + qi];
            lp = p.pre_l[head * p.Lq + qi];
        }
        const float mt = fmaxf(m, mp);
        const float a = __builtin_amdgcn_exp2f((m - mt) * 1.44269504088896340736f);
        const float bs = __builtin_amdgcn_exp2f((mp - mt) * 1.44269504088896340736f);
        const float lt = l * a + lp * bs;
        inv = a / lt;
        if (lh == 0) fbuf[wid * 32 + l31] = bs / lt;
    }

    // ---- O = P V
    f32x16 oacc[HDT];
#pragma unroll
    for (int d = 0; d < HDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    // transposed-read addressing: 16-lane group g reads the 4-key x 16-dim block at dims (g&1)*16..+15;
    // lane 4q+p of the group supplies row q, dims 4p..4p+3
    const int gi = lane & 15;
    const int tr_off = (gi >> 2) * ROWB + (((lane >> 4) & 1) * 16 + (gi & 3) * 4) * 2;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j0 = c * 64;
        if (j0 < Lk) {
            for (int x = tid; x < 64 * CPR; x += nthr) {
                const int r = x / CPR, kc = x % CPR, j = j0 + r;
                u32x4 v = zero4;
                if (j < p.L1)
                    v = *(const u32x4*)(V1 + (long long)j * p.ldkv1 + kc * 8);
                else if (j < Lk)
                    v = *(const u32x4*)(V2 + (long long)(j - p.L1) * p.ldkv2 + kc * 8);
                *(u32x4*)(lds + r * ROWB + kc * 16) = v;
            }
            __syncthreads();
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                if (j0 + jt * 32 < Lk) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 pa;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pa[e] = (bf16_t)(sacc[c][jt][8 * s2 + e] * inv);
                        const int kb = jt * 32 + 16 * s2 + 4 * lh;  // key rows kb..kb+3 and kb+8..kb+11
#pragma unroll
                        for (int d = 0; d < HDT; ++d) {
                            const char* base = lds + kb * ROWB + d * 64 + tr_off;
                            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (s16x4 __attribute__((address_space(3)))*)(base));
                            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (s16x4 __attribute__((address_space(3)))*)(base + 8 * ROWB));
                            const s16x8 vb = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, __builtin_bit_cast(bf16x8, vb), oacc[d], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- store O[i][dim]: row i = (e&3) + 8(e>>2) + 4h of this wave's 32 queries, dim = d*32 + l31
    bf16_t* Ob = (bf16_t*)p.O + b * p.o_bstride + head * HD;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = q0 + wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        int orow = -1;
        if (i < p.Lq)
            orow = p.orow1 + i;
        else if (Q2b && i >= Lq1p && i - Lq1p < p.Lq2)
            orow = p.orow2 + i - Lq1p;
        if (orow >= 0) {
            if (p.pre_m) {
                const float f = fbuf[i - q0];
                const float* po = p.pre_O + ((long long)head * p.Lq + i) * HD;
#pragma unroll
                for (int d = 0; d < HDT; ++d) oacc[d][e] = fmaf(f, po[d * 32 + l31], oacc[d][e]);
            }
#pragma unroll
            for (int d = 0; d < HDT; ++d) Ob[(long long)orow * p.ldo + d * 32 + l31] = (bf16_t)oacc[d][e];
        }
    }
}

// Lk <= 64 (every bf16 attention of the T = 32 plan step).  The kernel above stages Q, then K, then V through LDS with
// a barrier after each: three dependent global round trips per tiny block.  Here the Q and K fragments are read
// straight from global memory in MFMA operand order (lane (row, h) reads the 16 bytes holding dims 16s+8h..+7 of its
// row; all 8 + 16 loads of a wave are in flight together), only V goes through LDS (its B operand needs the
// transposing ds_read_b64_tr_b16), and V's loads are issued as soon as the K registers are free so that the softmax
// runs under them.  One wave per 32-query tile, NW tiles per block sharing the V image.  Arithmetic and its order are
// those of attn_bf16_kernel<HDT, 1>.
// NKT = 32-key tiles held in accumulators (2: Lk <= 64, 4: Lk <= 128)
template <int HDT, int NW, int NKT>
__global__ __launch_bounds__(NW * 64, 2) void attn_bf16_direct_kernel(AttnP p) {
    constexpr int HD = HDT * 32;
    constexpr int ROWB = HD * 2 + 16;
    constexpr int NS = HD / 16;
    constexpr int CPR = HD / 8;
    constexpr int NTHR = NW * 64;
    constexpr int VPT = NKT * 32 * CPR / NTHR;  // 16-byte V chunks per thread
    static_assert(NKT * 32 * CPR % NTHR == 0, "V image must divide over the block");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* fbuf = (float*)(lds + NKT * 32 * ROWB);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x, head = blockIdx.y, q0 = blockIdx.z * (NW * 32);
    const int Lk = p.L1 + p.L2;
    const int l31 = lane & 31, lh = lane >> 5;
    const bf16_t* Qb = (const bf16_t*)p.Q + b * p.q_bstride + head * HD;
    const bf16_t* K1 = (const bf16_t*)p.K1 + b * p.kv1_bstride + head * HD;
    const bf16_t* V1 = (const bf16_t*)p.V1 + b * p.kv1_bstride + head * HD;
    const bf16_t* K2 = p.K2 ? (const bf16_t*)p.K2 + head * HD : nullptr;
    const bf16_t* V2 = p.V2 ? (const bf16_t*)p.V2 + head * HD : nullptr;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int Lq1p = (p.Lq + 31) & ~31;
    const bf16_t* Q2b = p.Q2 ? (const bf16_t*)p.Q2 + head * HD : nullptr;

    // this lane's query row (query slot q0 + wid*32 + l31) and its two key rows (keys l31 and 32 + l31)
    const int qi = q0 + wid * 32 + l31;
    const bf16_t* qrow = nullptr;
    if (qi < p.Lq)
        qrow = Qb + (long long)qi * p.ldq;
    else if (Q2b && qi >= Lq1p && qi - Lq1p < p.Lq2)
        qrow = Q2b + (long long)(qi - Lq1p) * p.ldq2;
    u32x4 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        qf[s] = zero4;
        if (qrow) qf[s] = *(const u32x4*)(qrow + 16 * s + 8 * lh);
    }
    f32x16 sacc[NKT];
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[jt][e] = 0.f;
#pragma unroll
    for (int pr = 0; pr < NKT / 2; ++pr) {  // two key tiles' fragments in registers at a time
        u32x4 kf[2][NS];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = (2 * pr + t) * 32 + l31;
            const bf16_t* krow = j < p.L1 ? K1 + (long long)j * p.ldkv1 : (j < Lk ? K2 + (long long)(j - p.L1) * p.ldkv2 : nullptr);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                kf[t][s] = zero4;
                if (krow) kf[t][s] = *(const u32x4*)(krow + 16 * s + 8 * lh);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if ((2 * pr + t) * 32 < Lk) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    sacc[2 * pr + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[t][s]),
                                                                               __builtin_bit_cast(bf16x8, qf[s]), sacc[2 * pr + t], 0, 0, 0);
            }
    }
    // V: issue the loads now (the K registers are dead), park them in LDS after the softmax
    u32x4 vv[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int x = tid + i * NTHR, r = x / CPR, kc = x % CPR;
        vv[i] = zero4;
        if (r < p.L1)
            vv[i] = *(const u32x4*)(V1 + (long long)r * p.ldkv1 + kc * 8);
        else if (r < Lk)
            vv[i] = *(const u32x4*)(V2 + (long long)(r - p.L1) * p.ldkv2 + kc * 8);
    }

    float m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float v = (j < Lk) ? sacc[jt][e] * p.scale : -INFINITY;
            sacc[jt][e] = v;
            m = fmaxf(m, v);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = __builtin_amdgcn_exp2f((sacc[jt][e] - m) * 1.44269504088896340736f);
            sacc[jt][e] = v;
            l += v;
        }
    l += __shfl_xor(l, 32);
    float inv = 1.0f / l;
    if (p.pre_m) {
        float mp = -INFINITY, lp = 0.f;
        if (qi < p.Lq) {
            mp = p.pre_m[head * p.Lq + qi];
            lp = p.pre_l[head * p.Lq + qi];
        }
        const float mt = fmaxf(m, mp);
        const float a = __builtin_amdgcn_exp2f((m - mt) * 1.44269504088896340736f);
        const float bs = __builtin_amdgcn_exp2f((mp - mt) * 1.44269504088896340736f);
        const float lt = l * a + lp * bs;
        inv = a / lt;
        if (lh == 0) fbuf[wid * 32 + l31] = bs / lt;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int x = tid + i * NTHR, r = x / CPR, kc = x % CPR;
        *(u32x4*)(lds + r * ROWB + kc * 16) = vv[i];
    }
    __syncthreads();

    f32x16 oacc[HDT];
#pragma unroll
    for (int d = 0; d < HDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    const int gi = lane & 15;
    const int tr_off = (gi >> 2) * ROWB + (((lane >> 4) & 1) * 16 + (gi & 3) * 4) * 2;
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt) {
        if (jt * 32 < Lk) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pa;
#pragma unroll
                for (int e = 0; e < 8; ++e) pa[e] = (bf16_t)(sacc[jt][8 * s2 + e] * inv);
                const int kb = jt * 32 + 16 * s2 + 4 * lh;
#pragma unroll
                for (int d = 0; d < HDT; ++d) {
                    const char* base = lds + kb * ROWB + d * 64 + tr_off;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
                    const s16x4 v1 =
                        __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 8 * ROWB));
                    const s16x8 vb = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, __builtin_bit_cast(bf16x8, vb), oacc[d], 0, 0, 0);
                }
            }
        }
    }
    bf16_t* Ob = (bf16_t*)p.O + b * p.o_bstride + head * HD;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = q0 + wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        int orow = -1;
        if (i < p.Lq)
            orow = p.orow1 + i;
        else if (Q2b && i >= Lq1p && i - Lq1p < p.Lq2)
            orow = p.orow2 + i - Lq1p;
        if (orow >= 0) {
            if (p.pre_m) {
                const float f = fbuf[i - q0];
                const float* po = p.pre_O + ((long long)head * p.Lq + i) * HD;
#pragma unroll
                for (int d = 0; d < HDT; ++d) oacc[d][e] = fmaf(f, po[d * 32 + l31], oacc[d][e]);
            }
#pragma unroll
            for (int d = 0; d < HDT; ++d) Ob[(long long)orow * p.ldo + d * 32 + l31] = (bf16_t)oacc[d][e];
        }
    }
}
// Lk <= 52, at most 52 query slots, head dim 128 (the encoder attentions of the T = 32 plan step): the same arithmetic as
// attn_bf16_direct_kernel<4, 2, 2>, software-pipelined over (batch element, head) items.  The direct kernel is one
// short-lived workgroup per item: load -> compute -> store with nothing in flight during the last two, all workgroups of a
// round in the same stage at the same time (31.5 us for 100 MB where the loads alone take 17.5).  Here a workgroup is
// persistent, two per CU, and has THREE waves: two compute an item (32 query slots each) while the third, the loader,
// keeps the next two items' Q, K, V rows in flight -- LDS-DMA in whole-line pieces (4 rows x 256 B, 16-byte chunks
// swizzled by row & 7 on the source side so that the fragment reads are conflict-free) into two LDS buffers.  The split
// is about the counter: vector-memory operations complete in issue order, so a wave that both loads and stores waits for
// its old stores' acknowledgements whenever it waits for a load (measured: 42 us with the loads and stores on the same
// waves); the loader's vmcnt sees loads only, and the compute waves never wait for their stores.
// O leaves through the Q image of the item's buffer (a wave's Q rows are in registers by then; P V runs with the
// operands swapped -- the register images are identical -- so a lane owns its query row and writes 8 bytes per quarter
// of a 32-dim tile): whole 256-byte rows out, 4 rows per store instruction.
// Images per buffer: V | K | Q, 52 rows each.  Fragment rows past an image (keys or queries 52..63) read the image behind
// it: finite values that are masked (scores of keys >= Lk), multiplied by an exact zero (V rows of keys >= Lk: rows
// Lk..51 are never written and are zeroed once) or never stored (queries).
// NQ1, NQ2, NK1, NK2: 4-row pieces of the per-item and the batch-shared query / key segments.
namespace {
constexpr int APIPE_NR = 52, APIPE_IMG = APIPE_NR * 256, APIPE_BUF = 3 * APIPE_IMG;
}
template <int N1, int SH>
__global__ __launch_bounds__(192, 2) void attn_bf16_pipe_kernel(AttnP p, int n_items) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int HD = 128, NS = 8, HDT = 4;
    constexpr int Lk = N1 + SH;                                  // keys: the item's own N1 rows, then SH rows shared by the batch
    constexpr int NP_OWN = (N1 + 3) / 4, NP_SH = SH ? (Lk + 3) / 4 - N1 / 4 : 0;  // 4-row pieces of the two key segments (the one at the seam twice)
    constexpr int NDMA = NP_OWN + 2 * (NP_OWN + NP_SH);          // pieces per item (the loader's): Q, K, V
    static_assert(NDMA <= 63 && Lk <= APIPE_NR && (SH == 0 || (SH == 32 && N1 <= APIPE_NR - 32)), "shapes");
    extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 * APIPE_BUF bytes (dynamic: two workgroups per CU must fit to the byte)
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0, 1: compute; 2: loader
    const int l31 = lane & 31, lh = lane >> 5;
    // query slots: SH == 0: slot i = the item's query i.  SH == 32 (first layer): slots 0..N1-1 = the item's queries (wave 0),
    // slots 32..63 = the 32 shared ones (wave 1: the same rows for every item of a head, and a workgroup's items share the head)
    constexpr int QROW0 = SH ? 32 : 0;  // image row of the item's first query = its output row (run_block: shared rows first)

    {   // the V rows of keys >= Lk are never written and must be zero: P is exactly zero there, and 0 x (what LDS held) must not be a NaN
        const u32x4 z = {0u, 0u, 0u, 0u};
        constexpr int NZ = (APIPE_NR - Lk) * 16;  // 16-byte chunks per buffer
        for (int i = tid; i < 2 * NZ; i += 192) *(u32x4*)(lds + (i / NZ) * APIPE_BUF + Lk * 256 + (i % NZ) * 16) = z;
    }
    const int r4 = lane >> 4, c16 = lane & 15;  // a piece: row r4 of its 4, 16-byte chunk c16
    const int n_mine = ((int)blockIdx.x < n_items) ? (n_items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int stride = gridDim.x;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the zeros are in before a piece lands on them)

    if (wid == 2) {
        // ---------------------------------------------------------------- loader
        // this lane's source offset inside image piece pc (rows 4 pc .. 4 pc + 3, swizzled by the image row & 7 = 4 (pc & 1) + r4)
        auto voff = [&](int ld, int odd) { return (unsigned)(r4 * ld * 2 + ((c16 ^ (4 * odd + r4)) << 4)); };
        const unsigned vq[2] = {voff(p.ldq, 0), voff(p.ldq, 1)};
        const unsigned vk[2] = {voff(p.ldkv1, 0), voff(p.ldkv1, 1)}, vk2[2] = {voff(p.ldkv2, 0), voff(p.ldkv2, 1)};
        // source rows [0, L) of `src` (row stride ld elements) -> image rows R0 .. R0 + L - 1: the image pieces that hold them, lanes
        // of other rows switched off (a piece at the seam of two segments is issued once for each)
        auto seg_dma = [&](const bf16_t* src, int ld, const unsigned (&v)[2], char* img, auto r0_c, auto l_c, auto aux_c) {
            constexpr int R0 = decltype(r0_c)::value, L = decltype(l_c)::value, AUX = decltype(aux_c)::value;  // (AUX: the item's own rows are read once: streamed)
            if (L == 0) return;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src - (long long)R0 * ld), 0, (unsigned)((R0 + L) * ld * 2), 0x00020000);
#pragma unroll
            for (int pc = R0 / 4; pc <= (R0 + L - 1) / 4; ++pc) {
                const bool lo = 4 * pc >= R0 ? true : 4 * pc + r4 >= R0;
                const bool hi = 4 * pc + 3 < R0 + L ? true : 4 * pc + r4 < R0 + L;
                if (lo && hi) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(img + pc * 1024), 16, v[pc & 1], pc * 4 * ld * 2, 0, AUX);
            }
        };
        using I0 = std::integral_constant<int, 0>;
        auto issue = [&](int it, int buf) {
            if (p.no_pipe == 3) return;  // (lab timing: no loads)
            const int b = it >> 2, head = it & 3;  // (4 heads: try_pipe)
            char* const B = lds + buf * APIPE_BUF;
            using AS = std::integral_constant<int, M3PC_STREAM_AUX>;  // the item's own rows: read once; the shared segment: by every item of the head
            seg_dma((const bf16_t*)p.K1 + b * p.kv1_bstride + head * HD, p.ldkv1, vk, B + APIPE_IMG, I0{}, std::integral_constant<int, N1>{}, AS{});
            seg_dma((const bf16_t*)p.K2 + head * HD, p.ldkv2, vk2, B + APIPE_IMG, std::integral_constant<int, N1>{}, std::integral_constant<int, SH>{}, I0{});
            seg_dma((const bf16_t*)p.Q + b * p.q_bstride + head * HD, p.ldq, vq, B + 2 * APIPE_IMG, std::integral_constant<int, QROW0>{}, std::integral_constant<int, N1>{}, AS{});
            seg_dma((const bf16_t*)p.V1 + b * p.kv1_bstride + head * HD, p.ldkv1, vk, B, I0{}, std::integral_constant<int, N1>{}, AS{});
            seg_dma((const bf16_t*)p.V2 + head * HD, p.ldkv2, vk2, B, std::integral_constant<int, N1>{}, std::integral_constant<int, SH>{}, I0{});
        };
        if (n_mine > 0) issue(blockIdx.x, 0);
        if (n_mine > 1) issue(blockIdx.x + stride, 1);
        for (int k = 0; k < n_mine; ++k) {
            // item k is in (item k + 1 may be on its way: only at k = 0 -- later its pieces are issued behind barrier A, while the
            // compute waves work: issued in front of it, the 39 pieces' issue time sat between two items' arithmetic)
            if (k == 0 && n_mine > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");  // A: the compute waves start on item k
            if (k >= 1 && k + 1 < n_mine) issue(blockIdx.x + (k + 1) * stride, (k + 1) & 1);  // (its buffer is free since barrier B of item k - 1)
            asm volatile("s_barrier" ::: "memory");  // B: they are done with the buffer
        }
        return;
    }
    // -------------------------------------------------------------------- compute waves
    const int qi = wid * 32 + l31;  // this lane's query slot
    const int sw = l31 & 7;         // fragment addresses: chunk ^ (row & 7)
    const int gi = lane & 15;
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (lab: shader-clock stamps of this workgroup's third item)
    u32x4 qf[NS];
    if (SH && wid == 1) {  // the shared queries of this workgroup's head, once
        const bf16_t* qrow = (const bf16_t*)p.Q2 + (blockIdx.x % p.n_head) * HD + (long long)l31 * p.ldq2 + 8 * lh;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *(const u32x4*)(qrow + 16 * s);
    }
    for (int k = 0; k < n_mine; ++k) {
        const int it = blockIdx.x + k * stride;
        const char* const B = lds + (k & 1) * APIPE_BUF;
        asm volatile("s_barrier" ::: "memory");  // A
        if (p.stamps && k == 2) st_[0] = __builtin_readcyclecounter();
        if (p.no_pipe == 2) {  // (lab timing: no arithmetic)
            asm volatile("s_barrier" ::: "memory");
            continue;
        }
        // ---- S^T = K Q^T
        const unsigned vbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)B;  // (the V image: first in the buffer)
        const char* const Ki = B + APIPE_IMG;
        const char* const Qi = B + 2 * APIPE_IMG;
        if (SH == 0 || wid == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) qf[s] = *(const u32x4*)(Qi + (QROW0 + qi) * 256 + (((2 * s + lh) ^ sw) << 4));
        }
        f32x16 sacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)  // (the two key tiles' accumulator chains side by side)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 kf = *(const u32x4*)(Ki + (32 * t + l31) * 256 + (((2 * s + lh) ^ sw) << 4));
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), sacc[t], 0, 0, 0);
            }
        asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));
        if (p.stamps && k == 2) st_[1] = __builtin_readcyclecounter();
        // (a key slot that is past Lk in both lane halves is dead at compile time: its exp is an exact 0 in the sum and in P -- the
        // same bits as the direct kernel's exp2(-inf) -- and costs nothing)
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (jt * 32 + (e & 3) + 8 * (e >> 2) >= Lk) continue;
                const int j = jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 < Lk || j < Lk) ? sacc[jt][e] * p.scale : -INFINITY;
                sacc[jt][e] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (jt * 32 + (e & 3) + 8 * (e >> 2) >= Lk) {
                    sacc[jt][e] = 0.f;
                    continue;
                }
                const float v = __builtin_amdgcn_exp2f((sacc[jt][e] - m) * 1.44269504088896340736f);
                sacc[jt][e] = v;
                l += v;
            }
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        // ---- O^T = V^T P^T (lane = query, registers = dims)
        asm volatile("" : "+v"(l));
        if (p.stamps && k == 2) st_[2] = __builtin_readcyclecounter();
        f32x16 oacc[HDT];
#pragma unroll
        for (int d = 0; d < HDT; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
        // transposing reads of 16-key step n = 2 jt + s2: lane -> key row kr (and kr + 8), 8 bytes at dim d * 32 + ((lane >> 4) & 1) * 16 +
        // (gi & 3) * 4.  As asm statements (the builtin carries no memory operand, and hipcc would order it behind every LDS-DMA
        // it has seen with a full vmcnt(0)), one step ahead of the MFMAs that use them: LDS operations return in order, so
        // lgkmcnt(8) = "all but the 8 reads of the next step".
        s16x4 tv[2][2 * HDT];
        auto tr_reads = [&](int n, s16x4 (&v)[2 * HDT]) {
            const int kr = 16 * n + 4 * lh + (gi >> 2);
#pragma unroll
            for (int d = 0; d < HDT; ++d) {
                const int bo = d * 64 + ((lane >> 4) & 1) * 32 + (gi & 3) * 8;  // byte offset in the row
                const unsigned a0 = vbase + kr * 256 + ((((bo >> 4) ^ (kr & 7)) << 4) | (bo & 15));
                const unsigned a1 = vbase + (kr + 8) * 256 + ((((bo >> 4) ^ ((kr + 8) & 7)) << 4) | (bo & 15));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[2 * d]) : "v"(a0));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[2 * d + 1]) : "v"(a1));
            }
        };
        tr_reads(0, tv[0]);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int jt = n >> 1, s2 = n & 1;
            bf16x8 pa;
#pragma unroll
            for (int e = 0; e < 8; ++e) pa[e] = (bf16_t)(sacc[jt][8 * s2 + e] * inv);
            s16x4(&cur)[2 * HDT] = tv[n & 1];
            if (n < 3) {
                tr_reads(n + 1, tv[(n + 1) & 1]);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
            }
#pragma unroll
            for (int d = 0; d < HDT; ++d) {
                const s16x8 vb = __builtin_shufflevector(cur[2 * d], cur[2 * d + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vb), pa, oacc[d], 0, 0, 0);
            }
        }
        asm volatile("" : "+v"(oacc[0]), "+v"(oacc[1]), "+v"(oacc[2]), "+v"(oacc[3]));
        if (p.stamps && k == 2) st_[3] = __builtin_readcyclecounter();
        {
            char* const Oi = (char*)Qi;
            // (output row = image row: SH == 0: the query's slot; first layer: the shared queries' rows 0..31, then the item's own)
            const int orow_i = SH == 0 ? qi : (wid == 0 ? 32 + qi : qi - 32);
            if (SH == 0 ? qi < APIPE_NR : (wid == 1 || qi < N1)) {
#pragma unroll
                for (int d = 0; d < HDT; ++d)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x4 w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] = (bf16_t)oacc[d][4 * q + i];
                        *(bf16x4*)(Oi + orow_i * 256 + (((4 * d + q) ^ (orow_i & 7)) << 4) + 8 * lh) = w;
                    }
            }
        }
        if (p.stamps && k == 2) st_[4] = __builtin_readcyclecounter();
        // read back the rows this wave wrote (same wave: LDS operations execute in order, no barrier): 4 rows per piece and store
        constexpr int NPC_LO = 8, NPC_HI = APIPE_NR / 4 - 8;  // pieces of image rows 0..31 / 32..51
        const bool low_rows = SH == 0 ? wid == 0 : wid == 1;    // (first layer: wave 1 owns the shared queries = rows 0..31)
        u32x4 ov[NPC_LO];
#pragma unroll
        for (int j = 0; j < NPC_LO; ++j) {
            const int pc = low_rows ? j : (j < NPC_HI ? 8 + j : 8 + NPC_HI - 1);
            ov[j] = *(const u32x4*)(Qi + (4 * pc + r4) * 256 + c16 * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B: the pieces of the item after next may land
        if (p.stamps && k == 2) st_[5] = __builtin_readcyclecounter();
        // ---- stores (never waited for; rows that do not exist are addressed out of the buffer's range; the wave of the upper
        // rows repeats its last piece so that both issue the same number)
        {
            const int b = it >> 2, head = it & 3;  // (4 heads: try_pipe)
            const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((bf16_t*)p.O + b * p.o_bstride + head * HD), 0, 0x7fffffffu, 0x00020000);
#pragma unroll
            for (int j = 0; j < NPC_LO; ++j) {
                const int pc = low_rows ? j : (j < NPC_HI ? 8 + j : 8 + NPC_HI - 1);
                const int row = 4 * pc + r4;
                const unsigned off = row < Lk ? (unsigned)((SH ? row : p.orow1 + row) * p.ldo * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(ov[j], o_rs, off + ((c16 ^ (row & 7)) << 4), 0, M3PC_STREAM_AUX);
            }
        }
        if (p.stamps && k == 2) st_[6] = __builtin_readcyclecounter();
    }
    if (p.stamps && (int)blockIdx.x == 37 && lane == 0)
        for (int i = 0; i < 7; ++i) p.stamps[wid * 8 + i] = st_[i];
#endif
}
// The decoder's attention of an rtg_guiding candidate pass in the same pipelined form: 32 queries that are the SAME rows for every
// candidate (the masked scored tokens' query table) against the candidate's own N1 K|V rows, merged with the pre-reduced block of
// the masked tokens' keys (AttnP::pre_m / pre_l / pre_O) exactly as attn_bf16_direct_kernel<4, 1, 2> merges it.  One 32-query tile
// per item = one wave, so the workgroup's two compute waves work on two DIFFERENT items at a time (each with its own pair of
// K|V buffers: 4 x 26 KB, one workgroup per CU); a workgroup's items share the head, so the Q fragments, the pre-block's
// statistics and its pre_O rows are loaded once and stay in registers.  O leaves through the wave's own K image.
template <int N1>
__global__ __launch_bounds__(192, 1) void attn_bf16_pipe_dec_kernel(AttnP p, int n_items) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int HD = 128, NS = 8, HDT = 4;
    constexpr int Lk = N1;
    constexpr int NPC = (N1 + 3) / 4;       // 4-row pieces of one image
    constexpr int NDMA = 2 * 2 * NPC;       // pieces per round (two items' K and V), the loader's
    constexpr int DBUF = 2 * APIPE_IMG;     // one buffer: V | K
    static_assert(NDMA <= 63 && Lk <= APIPE_NR, "shapes");
    extern __shared__ __attribute__((aligned(16))) char lds[];  // 4 * DBUF bytes: wave w's buffers at (2 w + r % 2) * DBUF
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0, 1: compute; 2: loader
    const int l31 = lane & 31, lh = lane >> 5;
    {   // the V rows of keys >= Lk are never written and must be zero (P is exactly zero there)
        const u32x4 z = {0u, 0u, 0u, 0u};
        constexpr int NZ = (APIPE_NR - Lk) * 16;
        for (int i = tid; i < 4 * NZ; i += 192) *(u32x4*)(lds + (i / NZ) * DBUF + Lk * 256 + (i % NZ) * 16) = z;
    }
    const int r4 = lane >> 4, c16 = lane & 15;
    const int stride = gridDim.x;
    // this workgroup's items: blockIdx.x + stride * j; compute wave w takes j = 2 r + w in round r
    const int n_mine = ((int)blockIdx.x < n_items) ? (n_items - 1 - (int)blockIdx.x) / stride + 1 : 0;
    const int n_rounds = (n_mine + 1) / 2;
    const int head = blockIdx.x & 3;  // (4 heads, grid a multiple of 4: try_pipe)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    if (wid == 2) {
        // ---------------------------------------------------------------- loader
        const unsigned v[2] = {(unsigned)(r4 * p.ldkv1 * 2 + ((c16 ^ r4) << 4)), (unsigned)(r4 * p.ldkv1 * 2 + ((c16 ^ (4 + r4)) << 4))};
        auto issue = [&](int r) {
            if (p.no_pipe == 3) return;  // (lab timing: no loads)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                int j = 2 * r + w;
                j = j < n_mine ? j : n_mine - 1;  // (a wave without an item in the last round: its neighbour's rows again, never used)
                const int it = blockIdx.x + stride * j;
                char* const B = lds + (2 * w + (r & 1)) * DBUF;
                const bf16_t* kb = (const bf16_t*)p.K1 + (long long)(it >> 2) * p.kv1_bstride + head * HD;
                const bf16_t* vb = (const bf16_t*)p.V1 + (long long)(it >> 2) * p.kv1_bstride + head * HD;
                const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (unsigned)(N1 * p.ldkv1 * 2), 0x00020000);
                const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (unsigned)(N1 * p.ldkv1 * 2), 0x00020000);
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    const bool ok = 4 * pc + 3 < N1 ? true : 4 * pc + r4 < N1;
                    if (ok) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lptr_t)(B + APIPE_IMG + pc * 1024), 16, v[pc & 1], pc * 4 * p.ldkv1 * 2, 0, M3PC_STREAM_AUX);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lptr_t)(B + pc * 1024), 16, v[pc & 1], pc * 4 * p.ldkv1 * 2, 0, M3PC_STREAM_AUX);
                    }
                }
            }
        };
        if (n_rounds > 0) issue(0);
        if (n_rounds > 1) issue(1);
        for (int r = 0; r < n_rounds; ++r) {
            if (r == 0 && n_rounds > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");  // A
            if (r >= 1 && r + 1 < n_rounds) issue(r + 1);
            asm volatile("s_barrier" ::: "memory");  // B
        }
        return;
    }
    // -------------------------------------------------------------------- compute waves
    const int qi = l31;  // this lane's query
    const int sw = l31 & 7;
    const int gi = lane & 15;
    // per workgroup constants: the Q fragments of this head's 32 shared queries, the pre-reduced block's statistics and pre_O rows
    u32x4 qf[NS];
    {
        const bf16_t* qrow = (const bf16_t*)p.Q + head * HD + (long long)(qi < p.Lq ? qi : 0) * p.ldq + 8 * lh;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *(const u32x4*)(qrow + 16 * s);
    }
    float mp = -INFINITY, lp = 0.f;
    f32x4v po[HDT][4];  // pre_O of this lane's query at the dims it owns: d * 32 + 8 q + 4 lh ..
    if (qi < p.Lq) {
        mp = p.pre_m[head * p.Lq + qi];
        lp = p.pre_l[head * p.Lq + qi];
    }
    {
        const float* prow = p.pre_O + ((long long)head * p.Lq + (qi < p.Lq ? qi : 0)) * HD + 4 * lh;
#pragma unroll
        for (int d = 0; d < HDT; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) po[d][q] = *(const f32x4v*)(prow + d * 32 + 8 * q);
    }
    const unsigned ooff_base = qi < p.Lq ? 0u : 0x80000000u;
    (void)ooff_base;
    for (int r = 0; r < n_rounds; ++r) {
        const int j = 2 * r + wid;
        const bool have = j < n_mine;
        const int it = blockIdx.x + stride * (have ? j : 0);
        const char* const B = lds + (2 * wid + (r & 1)) * DBUF;
        asm volatile("s_barrier" ::: "memory");  // A
        if (have && p.no_pipe != 2) {
            const unsigned vbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)B;
            const char* const Ki = B + APIPE_IMG;
            f32x16 sacc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const u32x4 kf = *(const u32x4*)(Ki + (32 * t + l31) * 256 + (((2 * s + lh) ^ sw) << 4));
                    sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), sacc[t], 0, 0, 0);
                }
            float m = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (jt * 32 + (e & 3) + 8 * (e >> 2) >= Lk) continue;
                    const int jj = jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float v = (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 < Lk || jj < Lk) ? sacc[jt][e] * p.scale : -INFINITY;
                    sacc[jt][e] = v;
                    m = fmaxf(m, v);
                }
            m = fmaxf(m, __shfl_xor(m, 32));
            float l = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (jt * 32 + (e & 3) + 8 * (e >> 2) >= Lk) {
                        sacc[jt][e] = 0.f;
                        continue;
                    }
                    const float v = __builtin_amdgcn_exp2f((sacc[jt][e] - m) * 1.44269504088896340736f);
                    sacc[jt][e] = v;
                    l += v;
                }
            l += __shfl_xor(l, 32);
            // the pre-reduced block: two blocks of a streaming softmax (attn_bf16_direct_kernel)
            const float mt = fmaxf(m, mp);
            const float a = __builtin_amdgcn_exp2f((m - mt) * 1.44269504088896340736f);
            const float bs = __builtin_amdgcn_exp2f((mp - mt) * 1.44269504088896340736f);
            const float lt = l * a + lp * bs;
            const float inv = a / lt;
            const float fpre = bs / lt;
            f32x16 oacc[HDT];
#pragma unroll
            for (int d = 0; d < HDT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
            s16x4 tv[2][2 * HDT];
            auto tr_reads = [&](int n, s16x4 (&v)[2 * HDT]) {
                const int kr = 16 * n + 4 * lh + (gi >> 2);
#pragma unroll
                for (int d = 0; d < HDT; ++d) {
                    const int bo = d * 64 + ((lane >> 4) & 1) * 32 + (gi & 3) * 8;
                    const unsigned a0 = vbase + kr * 256 + ((((bo >> 4) ^ (kr & 7)) << 4) | (bo & 15));
                    const unsigned a1 = vbase + (kr + 8) * 256 + ((((bo >> 4) ^ ((kr + 8) & 7)) << 4) | (bo & 15));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[2 * d]) : "v"(a0));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[2 * d + 1]) : "v"(a1));
                }
            };
            tr_reads(0, tv[0]);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int jt = n >> 1, s2 = n & 1;
                bf16x8 pa;
#pragma unroll
                for (int e = 0; e < 8; ++e) pa[e] = (bf16_t)(sacc[jt][8 * s2 + e] * inv);
                s16x4(&cur)[2 * HDT] = tv[n & 1];
                if (n < 3) {
                    tr_reads(n + 1, tv[(n + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                }
#pragma unroll
                for (int d = 0; d < HDT; ++d) {
                    const s16x8 vb = __builtin_shufflevector(cur[2 * d], cur[2 * d + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vb), pa, oacc[d], 0, 0, 0);
                }
            }
            // ---- + the pre-reduced block, bf16, through this wave's K image (its scores are done), whole rows out
            {
                char* const Oi = (char*)Ki;
#pragma unroll
                for (int d = 0; d < HDT; ++d)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x4 w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] = (bf16_t)fmaf(fpre, po[d][q][i], oacc[d][4 * q + i]);
                        *(bf16x4*)(Oi + qi * 256 + (((4 * d + q) ^ sw) << 4) + 8 * lh) = w;
                    }
            }
            u32x4 ov[8];
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) ov[pc] = *(const u32x4*)(Ki + (4 * pc + r4) * 256 + c16 * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)((bf16_t*)p.O + (long long)(it >> 2) * p.o_bstride + head * HD), 0, 0x7fffffffu, 0x00020000);
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) {
                    const int row = 4 * pc + r4;
                    const unsigned off = row < p.Lq ? (unsigned)((p.orow1 + row) * p.ldo * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(ov[pc], o_rs, off + ((c16 ^ (row & 7)) << 4), 0, M3PC_STREAM_AUX);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B
    }
#endif
}
// The decoder's attention of a critic_lambda_guiding candidate pass (pruned_decoder with a prefix of un-masked queries): NQ1 queries of
// the candidate's own + the rest of a 32-query tile shared by the batch, against the candidate's own N1 K|V rows followed by N2 rows
// shared by the batch (the masked tokens').  The direct kernel gives the NQ1 own queries a 32-query tile of their own and re-reads the
// shared keys for every (candidate, head): 172 us per C3 half.  Here the own and the shared queries share ONE tile (a query's
// arithmetic does not depend on its slot), the shared K|V rows are fetched once per workgroup and stay in LDS, and the rest is the
// pipelined decoder kernel above: a loader wave, two compute waves on two different items.  Key order (own rows, then shared) and every
// sum's order are the direct kernel's: bit-identical.
template <int N1, int N2>
__global__ __launch_bounds__(192, 1) void attn_bf16_pipe_mix_kernel(AttnP p, int n_items) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int HD = 128, NS = 8, HDT = 4;
    constexpr int Lk = N1 + N2, NKT = (Lk + 31) / 32;
    constexpr int NPC1 = (N1 + 3) / 4, NPC2 = (N2 + 3) / 4;
    constexpr int NDMA = 2 * (2 * NPC1 + 1);          // pieces per round: two items' own K and V rows and own query rows
    constexpr int SIMG = NPC2 * 1024;                 // one shared image (K or V rows of the batch-shared keys)
    constexpr int OBUF = 2 * APIPE_IMG + 1024;        // one item buffer: V own | K own | Q own (one piece)
    static_assert(NDMA <= 63 && N1 <= APIPE_NR && NKT <= 4 && N1 >= 32, "shapes");
    extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 * SIMG + 4 * OBUF bytes
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    char* const Ks = lds;
    char* const Vs = lds + SIMG;
    char* const IB = lds + 2 * SIMG;
    {   // the shared V rows past N2 (the last piece's tail) must be zero: P is exactly zero there
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = tid; i < (NPC2 * 4 - N2) * 16; i += 192) *(u32x4*)(Vs + N2 * 256 + i * 16) = z;
    }
    const int r4 = lane >> 4, c16 = lane & 15;
    const int stride = gridDim.x;
    const int n_mine = ((int)blockIdx.x < n_items) ? (n_items - 1 - (int)blockIdx.x) / stride + 1 : 0;
    const int n_rounds = (n_mine + 1) / 2;
    const int head = blockIdx.x & 3;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    if (wid == 2) {
        // ---------------------------------------------------------------- loader
        auto voff = [&](int ld, int odd) { return (unsigned)(r4 * ld * 2 + ((c16 ^ (4 * odd + r4)) << 4)); };
        const unsigned v1[2] = {voff(p.ldkv1, 0), voff(p.ldkv1, 1)}, v2[2] = {voff(p.ldkv2, 0), voff(p.ldkv2, 1)};
        const unsigned vq = (unsigned)(r4 * p.ldq * 2 + ((c16 ^ r4) << 4));
        {   // the shared K|V rows, once
            const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.K2 + head * HD), 0, (unsigned)(N2 * p.ldkv2 * 2), 0x00020000);
            const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.V2 + head * HD), 0, (unsigned)(N2 * p.ldkv2 * 2), 0x00020000);
#pragma unroll
            for (int pc = 0; pc < NPC2; ++pc) {
                const bool ok = 4 * pc + 3 < N2 ? true : 4 * pc + r4 < N2;
                if (ok) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lptr_t)(Ks + pc * 1024), 16, v2[pc & 1], pc * 4 * p.ldkv2 * 2, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lptr_t)(Vs + pc * 1024), 16, v2[pc & 1], pc * 4 * p.ldkv2 * 2, 0, 0);
                }
            }
        }
        auto issue = [&](int r) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                int j = 2 * r + w;
                j = j < n_mine ? j : n_mine - 1;
                const int it = blockIdx.x + stride * j;
                char* const B = IB + (2 * w + (r & 1)) * OBUF;
                const bf16_t* kb = (const bf16_t*)p.K1 + (long long)(it >> 2) * p.kv1_bstride + head * HD;
                const bf16_t* vb = (const bf16_t*)p.V1 + (long long)(it >> 2) * p.kv1_bstride + head * HD;
                const bf16_t* qb = (const bf16_t*)p.Q + (long long)(it >> 2) * p.q_bstride + head * HD;
                const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (unsigned)(N1 * p.ldkv1 * 2), 0x00020000);
                const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (unsigned)(N1 * p.ldkv1 * 2), 0x00020000);
                const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, (unsigned)(p.Lq * p.ldq * 2), 0x00020000);
#pragma unroll
                for (int pc = 0; pc < NPC1; ++pc) {
                    const bool ok = 4 * pc + 3 < N1 ? true : 4 * pc + r4 < N1;
                    if (ok) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lptr_t)(B + APIPE_IMG + pc * 1024), 16, v1[pc & 1], pc * 4 * p.ldkv1 * 2, 0, M3PC_STREAM_AUX);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lptr_t)(B + pc * 1024), 16, v1[pc & 1], pc * 4 * p.ldkv1 * 2, 0, M3PC_STREAM_AUX);
                    }
                }
                if (r4 < p.Lq) __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (lptr_t)(B + 2 * APIPE_IMG), 16, vq, 0, 0, M3PC_STREAM_AUX);  // (Lq <= 4: one piece)
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");  // P: the shared rows are in
        if (n_rounds > 0) issue(0);
        if (n_rounds > 1) issue(1);
        for (int r = 0; r < n_rounds; ++r) {
            if (r == 0 && n_rounds > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");  // A
            if (r >= 1 && r + 1 < n_rounds) issue(r + 1);
            asm volatile("s_barrier" ::: "memory");  // B
        }
        return;
    }
    // -------------------------------------------------------------------- compute waves
    // query slots: 0 .. Lq-1 the item's own queries, Lq .. Lq + Lq2 - 1 the shared ones (their fragments: once)
    const int qi = l31;
    const int sw = l31 & 7;
    const int gi = lane & 15;
    const bool own_q = qi < p.Lq;
    u32x4 qsh[NS];
    {
        const int sq = qi - p.Lq;
        const bf16_t* qrow = (const bf16_t*)p.Q2 + head * HD + (long long)(sq >= 0 && sq < p.Lq2 ? sq : 0) * p.ldq2 + 8 * lh;
#pragma unroll
        for (int s = 0; s < NS; ++s) qsh[s] = *(const u32x4*)(qrow + 16 * s);
    }
    asm volatile("s_barrier" ::: "memory");  // P
    const unsigned ks_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)Ks;
    const unsigned vs_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)Vs;
    for (int r = 0; r < n_rounds; ++r) {
        const int j = 2 * r + wid;
        const bool have = j < n_mine;
        const int it = blockIdx.x + stride * (have ? j : 0);
        const char* const B = IB + (2 * wid + (r & 1)) * OBUF;
        asm volatile("s_barrier" ::: "memory");  // A
        if (have) {
            const unsigned vo_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)B;
            const unsigned ko_base = vo_base + APIPE_IMG;
            const char* const Qi = B + 2 * APIPE_IMG;
            u32x4 qf[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const u32x4 own = *(const u32x4*)(Qi + (own_q ? qi : 0) * 256 + (((2 * s + lh) ^ (own_q ? sw : 0)) << 4));
                qf[s] = own_q ? own : qsh[s];
            }
            f32x16 sacc[NKT];
#pragma unroll
            for (int t = 0; t < NKT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
            // (attn_bf16_direct_kernel<4, 2, 4>: two key tiles at a time)
#pragma unroll
            for (int pr = 0; pr < (NKT + 1) / 2; ++pr)
#pragma unroll
                for (int t = 2 * pr; t < 2 * pr + 2 && t < NKT; ++t) {
                    const int kj = 32 * t + l31;  // key: own row kj, or shared row kj - N1
                    const unsigned ka = 32 * t + 31 < N1 ? ko_base + kj * 256
                                        : (32 * t >= N1 ? ks_base + (kj - N1) * 256 : (kj < N1 ? ko_base + kj * 256 : ks_base + (kj - N1) * 256));
                    const int krow = 32 * t + 31 < N1 ? kj : (32 * t >= N1 ? kj - N1 : (kj < N1 ? kj : kj - N1));
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const u32x4 kf = *(const u32x4 __attribute__((address_space(3)))*)(uintptr_t)(ka + (((2 * s + lh) ^ (krow & 7)) << 4));
                        sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), sacc[t], 0, 0, 0);
                    }
                }
            float m = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int jj = jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float v = (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 < Lk || jj < Lk) ? sacc[jt][e] * p.scale : -INFINITY;
                    sacc[jt][e] = v;
                    m = fmaxf(m, v);
                }
            m = fmaxf(m, __shfl_xor(m, 32));
            float l = 0.f;
#pragma unroll
            for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = __builtin_amdgcn_exp2f((sacc[jt][e] - m) * 1.44269504088896340736f);
                    sacc[jt][e] = v;
                    l += v;
                }
            l += __shfl_xor(l, 32);
            const float inv = 1.0f / l;
            f32x16 oacc[HDT];
#pragma unroll
            for (int d = 0; d < HDT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
            s16x4 tv[2][2 * HDT];
            auto tr_reads = [&](int n, s16x4 (&v)[2 * HDT]) {
                const int kr = 16 * n + 4 * lh + (gi >> 2);  // key rows kr and kr + 8: own (< N1) or shared
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int k0 = kr + 8 * hf;
                    const bool own = 16 * n + 15 < N1 ? true : (16 * n >= N1 ? false : k0 < N1);
                    const int row = own ? k0 : k0 - N1;
                    const unsigned base = (own ? vo_base : vs_base) + row * 256;
#pragma unroll
                    for (int d = 0; d < HDT; ++d) {
                        const int bo = d * 64 + ((lane >> 4) & 1) * 32 + (gi & 3) * 8;
                        const unsigned a = base + ((((bo >> 4) ^ (row & 7)) << 4) | (bo & 15));
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[2 * d + hf]) : "v"(a));
                    }
                }
            };
            constexpr int NSTEP = 2 * NKT;
            tr_reads(0, tv[0]);
#pragma unroll
            for (int n = 0; n < NSTEP; ++n) {
                const int jt = n >> 1, s2 = n & 1;
                bf16x8 pa;
#pragma unroll
                for (int e = 0; e < 8; ++e) pa[e] = (bf16_t)(sacc[jt][8 * s2 + e] * inv);
                s16x4(&cur)[2 * HDT] = tv[n & 1];
                if (n + 1 < NSTEP) {
                    tr_reads(n + 1, tv[(n + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                }
#pragma unroll
                for (int d = 0; d < HDT; ++d) {
                    const s16x8 vb = __builtin_shufflevector(cur[2 * d], cur[2 * d + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vb), pa, oacc[d], 0, 0, 0);
                }
            }
            // ---- bf16, through this wave's own K image (its scores are done), whole rows out: output row = query slot
            {
                char* const Oi = (char*)(B + APIPE_IMG);
#pragma unroll
                for (int d = 0; d < HDT; ++d)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x4 w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] = (bf16_t)oacc[d][4 * q + i];
                        *(bf16x4*)(Oi + qi * 256 + (((4 * d + q) ^ sw) << 4) + 8 * lh) = w;
                    }
            }
            u32x4 ov[8];
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) ov[pc] = *(const u32x4*)(B + APIPE_IMG + (4 * pc + r4) * 256 + c16 * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)((bf16_t*)p.O + (long long)(it >> 2) * p.o_bstride + head * HD), 0, 0x7fffffffu, 0x00020000);
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) {
                    const int row = 4 * pc + r4;
                    const unsigned off = row < p.Lq + p.Lq2 ? (unsigned)(row * p.ldo * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(ov[pc], o_rs, off + ((c16 ^ (row & 7)) << 4), 0, M3PC_STREAM_AUX);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B
    }
#endif
}
// The T = 64 candidate pass (BASELINE config 4: 97 rows per candidate -- 33 own + 64 history rows in the first layer) in the
// pipelined form.  attn_bf16_direct_kernel<4, 2, 4> gives every (candidate, head, 64 query slots) a short-lived workgroup: 140-340 us
// per launch for 200-400 MB.  Same scheme as attn_bf16_pipe_kernel -- persistent workgroup, a loader wave that keeps the next item's rows
// in flight by LDS-DMA, compute waves that never wait for their stores, O out through the Q image in whole 256-byte rows -- cut for
// 100-row images: one workgroup per CU (2 x {V | K | Q} = 150 KB), one compute wave per 32-query tile of the item (up to 4), four
// 32-key tiles in the accumulators, seven 16-key steps of P V.  Query / output rows: the SHQ batch-shared ones first (their fragments
// stay in registers: a workgroup's items share the head), then the item's own NQ1; keys: the item's own NK1, then SHK shared.
// PRE: the decoder (all queries shared, the item's own K|V rows, merged with the pre-reduced block of the masked tokens' keys as
// attn_bf16_pipe_dec_kernel merges it).  Arithmetic and its order are the direct kernel's: bit-identical.
namespace {
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): loop indices as immediates
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
constexpr int WPIPE_NR = 100, WPIPE_IMG = WPIPE_NR * 256, WPIPE_BUF = 3 * WPIPE_IMG, WPIPE_LDS = 160 * 1024;
}
template <int NQ1, int SHQ, int NK1, int SHK, bool PRE>
__global__ __launch_bounds__(((NQ1 + SHQ + 31) / 32 + 1) * 64, 1) void attn_bf16_pipe_wide_kernel(AttnP p, int n_items) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int HD = 128, NS = 8, HDT = 4;
    constexpr int Lq = NQ1 + SHQ, Lk = NK1 + SHK;
    constexpr int NQT = (Lq + 31) / 32, NKT = (Lk + 31) / 32, NSTEP = (Lk + 15) / 16;
    constexpr int NTHR = (NQT + 1) * 64;
    constexpr int NPQ = NQ1 ? (Lq + 3) / 4 - SHQ / 4 : 0;                      // 4-row pieces: the own query rows,
    constexpr int NPK = (NK1 + 3) / 4 + (SHK ? (Lk + 3) / 4 - NK1 / 4 : 0);    // the two key segments (the piece at the seam twice)
    constexpr int NDMA = NPQ + 2 * NPK;
    static_assert(Lq <= WPIPE_NR && Lk <= WPIPE_NR && SHQ % 32 == 0 && NKT <= 4 && NQT <= 4 && (!PRE || (NQ1 == 0 && SHK == 0)), "shapes");
    // (query rows 100..127 of the second buffer's Q image read up to byte 2 * WPIPE_BUF - WPIPE_IMG + 128 * 256 of the allocation)
    static_assert(2 * WPIPE_BUF - WPIPE_IMG + 128 * 256 <= WPIPE_LDS, "LDS");
    extern __shared__ __attribute__((aligned(16))) char lds[];  // WPIPE_LDS bytes
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0 .. NQT - 1: compute (query rows 32 wid ..); NQT: loader
    const int l31 = lane & 31, lh = lane >> 5;
    {   // the V rows of keys >= Lk are never written and must be zero: P is exactly zero there
        const u32x4 z = {0u, 0u, 0u, 0u};
        constexpr int NZ = (WPIPE_NR - Lk) * 16;
        for (int i = tid; i < 2 * NZ; i += NTHR) *(u32x4*)(lds + (i / NZ) * WPIPE_BUF + Lk * 256 + (i % NZ) * 16) = z;
    }
    const int r4 = lane >> 4, c16 = lane & 15;
    const int n_mine = ((int)blockIdx.x < n_items) ? (n_items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int stride = gridDim.x;
    const int head = blockIdx.x & 3;  // (4 heads, grid a multiple of 4: try_pipe)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    if (wid == NQT) {
        // ---------------------------------------------------------------- loader
        auto voff = [&](int ld, int odd) { return (unsigned)(r4 * ld * 2 + ((c16 ^ (4 * odd + r4)) << 4)); };
        const unsigned vq[2] = {voff(p.ldq, 0), voff(p.ldq, 1)};
        const unsigned vk[2] = {voff(p.ldkv1, 0), voff(p.ldkv1, 1)}, vk2[2] = {voff(p.ldkv2, 0), voff(p.ldkv2, 1)};
        auto seg_dma = [&](const bf16_t* src, int ld, const unsigned (&v)[2], char* img, auto r0_c, auto l_c) {
            constexpr int R0 = decltype(r0_c)::value, L = decltype(l_c)::value;
            if (L == 0) return;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src - (long long)R0 * ld), 0, (unsigned)((R0 + L) * ld * 2), 0x00020000);
#pragma unroll
            for (int pc = R0 / 4; pc <= (R0 + L - 1) / 4; ++pc) {
                const bool lo = 4 * pc >= R0 ? true : 4 * pc + r4 >= R0;
                const bool hi = 4 * pc + 3 < R0 + L ? true : 4 * pc + r4 < R0 + L;
                if (lo && hi) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(img + pc * 1024), 16, v[pc & 1], pc * 4 * ld * 2, 0, 0);
            }
        };
        using I0 = std::integral_constant<int, 0>;
        auto issue = [&](int it, int buf) {
            if (p.no_pipe == 3) return;  // (lab timing: no loads)
            const long long b = it >> 2;
            char* const B = lds + buf * WPIPE_BUF;
            seg_dma((const bf16_t*)p.K1 + b * p.kv1_bstride + head * HD, p.ldkv1, vk, B + WPIPE_IMG, I0{}, std::integral_constant<int, NK1>{});
            seg_dma((const bf16_t*)p.K2 + head * HD, p.ldkv2, vk2, B + WPIPE_IMG, std::integral_constant<int, NK1>{}, std::integral_constant<int, SHK>{});
            seg_dma((const bf16_t*)p.Q + b * p.q_bstride + head * HD, p.ldq, vq, B + 2 * WPIPE_IMG, std::integral_constant<int, SHQ>{}, std::integral_constant<int, NQ1>{});
            seg_dma((const bf16_t*)p.V1 + b * p.kv1_bstride + head * HD, p.ldkv1, vk, B, I0{}, std::integral_constant<int, NK1>{});
            seg_dma((const bf16_t*)p.V2 + head * HD, p.ldkv2, vk2, B, std::integral_constant<int, NK1>{}, std::integral_constant<int, SHK>{});
        };
        if (n_mine > 0) issue(blockIdx.x, 0);
        if (n_mine > 1) issue(blockIdx.x + stride, 1);
        for (int k = 0; k < n_mine; ++k) {
            // (the counter holds 63: with more pieces per item the issue of item 1 already waited for item 0's oldest, and "63 outstanding"
            // of 2 NDMA still means item 0 is in)
            if (k == 0 && n_mine > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA < 63 ? NDMA : 63) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");  // A: the compute waves start on item k
            if (k >= 1 && k + 1 < n_mine) issue(blockIdx.x + (k + 1) * stride, (k + 1) & 1);
            asm volatile("s_barrier" ::: "memory");  // B: they are done with the buffer
        }
        return;
    }
    // -------------------------------------------------------------------- compute waves
    const int qrow_i = wid * 32 + l31;  // this lane's query = image row = output row (shared rows first)
    const int sw = l31 & 7;
    const int gi = lane & 15;
    const bool shared_q = wid * 32 < SHQ;  // (wave-uniform)
    u32x4 qf[NS];
    if (shared_q) {  // the shared queries of this workgroup's head, once
        const bf16_t* qsrc = PRE ? (const bf16_t*)p.Q : (const bf16_t*)p.Q2;
        const int ld = PRE ? p.ldq : p.ldq2;
        const int qlim = PRE ? p.Lq : SHQ;
        const bf16_t* qrow = qsrc + head * HD + (long long)(qrow_i < qlim ? qrow_i : 0) * ld + 8 * lh;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *(const u32x4*)(qrow + 16 * s);
    }
    float mp = -INFINITY, lp = 0.f;
    f32x4v po[PRE ? HDT : 1][4];  // (PRE) pre_O of this lane's query at the dims it owns: d * 32 + 8 q + 4 lh ..
    if constexpr (PRE) {
        if (qrow_i < p.Lq) {
            mp = p.pre_m[head * p.Lq + qrow_i];
            lp = p.pre_l[head * p.Lq + qrow_i];
        }
        const float* prow = p.pre_O + ((long long)head * p.Lq + (qrow_i < p.Lq ? qrow_i : 0)) * HD + 4 * lh;
#pragma unroll
        for (int d = 0; d < HDT; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) po[d][q] = *(const f32x4v*)(prow + d * 32 + 8 * q);
    }
    const int n_out = PRE ? p.Lq : Lq;  // output rows that exist
    for (int k = 0; k < n_mine; ++k) {
        const int it = blockIdx.x + k * stride;
        const char* const B = lds + (k & 1) * WPIPE_BUF;
        asm volatile("s_barrier" ::: "memory");  // A
        if (p.no_pipe == 2) {  // (lab timing: no arithmetic)
            asm volatile("s_barrier" ::: "memory");
            continue;
        }
        const unsigned vbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)B;  // (the V image: first in the buffer)
        const char* const Ki = B + WPIPE_IMG;
        const char* const Qi = B + 2 * WPIPE_IMG;
        if (!shared_q) {
#pragma unroll
            for (int s = 0; s < NS; ++s) qf[s] = *(const u32x4*)(Qi + qrow_i * 256 + (((2 * s + lh) ^ sw) << 4));
        }
        // ---- S^T = K Q^T (the key tiles' accumulator chains side by side)
        f32x16 sacc[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                const u32x4 kf = *(const u32x4*)(Ki + (32 * t + l31) * 256 + (((2 * s + lh) ^ sw) << 4));
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), sacc[t], 0, 0, 0);
            }
        // (a key slot past Lk in both lane halves is dead at compile time: an exact 0 in the sum and in P, the direct kernel's exp2(-inf))
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (jt * 32 + (e & 3) + 8 * (e >> 2) >= Lk) continue;
                const int j = jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 < Lk || j < Lk) ? sacc[jt][e] * p.scale : -INFINITY;
                sacc[jt][e] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (jt * 32 + (e & 3) + 8 * (e >> 2) >= Lk) {
                    sacc[jt][e] = 0.f;
                    continue;
                }
                const float v = __builtin_amdgcn_exp2f((sacc[jt][e] - m) * 1.44269504088896340736f);
                sacc[jt][e] = v;
                l += v;
            }
        l += __shfl_xor(l, 32);
        float inv = 1.0f / l, fpre = 0.f;
        if constexpr (PRE) {  // the pre-reduced block: two blocks of a streaming softmax (attn_bf16_direct_kernel)
            const float mt = fmaxf(m, mp);
            const float a = __builtin_amdgcn_exp2f((m - mt) * 1.44269504088896340736f);
            const float bs = __builtin_amdgcn_exp2f((mp - mt) * 1.44269504088896340736f);
            const float lt = l * a + lp * bs;
            inv = a / lt;
            fpre = bs / lt;
        }
        // ---- O^T = V^T P^T (lane = query, registers = dims).  P leaves the score accumulators as bf16 operands first (a wave has 256
        // registers here -- five waves on four SIMDs -- and 64 of scores + 64 of output + the transposed V reads do not fit beside the rest)
        u32x4 pa[NSTEP];
#pragma unroll
        for (int n = 0; n < NSTEP; ++n) {
            bf16x8 w;
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (bf16_t)(sacc[n >> 1][8 * (n & 1) + e] * inv);
            pa[n] = __builtin_bit_cast(u32x4, w);
            asm volatile("" : "+v"(pa[n]));
        }
        f32x16 oacc[HDT];
#pragma unroll
        for (int d = 0; d < HDT; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
        // transposing reads of 16-key step n (see attn_bf16_pipe_kernel; rows past the V image: K rows x exact zeros).  A step's 8 addresses
        // are 4 per-lane bases (one per 32-dim tile: the swizzle is by row & 7, which a 16-row step does not change) + immediates -- as
        // registers the 56 of them would be hoisted out of the item loop and spilled
        s16x4 tv[2][2 * HDT];
        unsigned tb[HDT];
        {
            const int kr = 4 * lh + (gi >> 2);
#pragma unroll
            for (int d = 0; d < HDT; ++d) {
                const int bo = d * 64 + ((lane >> 4) & 1) * 32 + (gi & 3) * 8;
                tb[d] = vbase + kr * 256 + ((((bo >> 4) ^ (kr & 7)) << 4) | (bo & 15));
            }
        }
        auto tr_reads = [](auto n_c, const unsigned (&tb)[HDT], s16x4 (&v)[2 * HDT]) {  // (no captures: hipcc rejects one in an asm operand of a generic lambda)
            constexpr int n = decltype(n_c)::value;
#pragma unroll
            for (int d = 0; d < HDT; ++d) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[2 * d]) : "v"(tb[d]), "n"(n * 4096));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[2 * d + 1]) : "v"(tb[d]), "n"(n * 4096 + 2048));
            }
        };
        tr_reads(std::integral_constant<int, 0>{}, tb, tv[0]);
        static_for<NSTEP>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            s16x4(&cur)[2 * HDT] = tv[n & 1];
            if constexpr (n < NSTEP - 1) {
                tr_reads(std::integral_constant<int, n + 1>{}, tb, tv[(n + 1) & 1]);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
            }
#pragma unroll
            for (int d = 0; d < HDT; ++d) {
                const s16x8 vb = __builtin_shufflevector(cur[2 * d], cur[2 * d + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vb), __builtin_bit_cast(bf16x8, pa[n]), oacc[d], 0, 0, 0);
            }
        });
        // ---- bf16 through this wave's rows of the Q image (its Q fragments are in registers), whole rows out
        {
            char* const Oi = (char*)Qi;
            if (qrow_i < WPIPE_NR) {
#pragma unroll
                for (int d = 0; d < HDT; ++d)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x4 w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] = (bf16_t)(PRE ? fmaf(fpre, po[PRE ? d : 0][q][i], oacc[d][4 * q + i]) : oacc[d][4 * q + i]);
                        *(bf16x4*)(Oi + qrow_i * 256 + (((4 * d + q) ^ sw) << 4) + 8 * lh) = w;
                    }
            }
        }
        // read back the rows this wave wrote (same wave: LDS operations execute in order), 4 rows per piece
        // (a wave whose tile ends before its 8 pieces -- the last one: rows 96..99 -- repeats the image's last piece and sends the repeats
        // out of the buffer's range: no branches, the same instruction stream for every wave)
        u32x4 ov[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pc = 8 * wid + j < WPIPE_NR / 4 ? 8 * wid + j : WPIPE_NR / 4 - 1;
            ov[j] = *(const u32x4*)(Qi + (4 * pc + r4) * 256 + c16 * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B: the pieces of the item after next may land
        {
            const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((bf16_t*)p.O + (long long)(it >> 2) * p.o_bstride + head * HD), 0, 0x7fffffffu, 0x00020000);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pc = 8 * wid + j;
                const int row = 4 * pc + r4;
                // (output row: with a shared segment the image row itself -- run_block: shared rows first; else orow1 + row)
                const unsigned off = pc < WPIPE_NR / 4 && row < n_out ? (unsigned)((SHQ && !PRE ? row : p.orow1 + row) * p.ldo * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(ov[j], o_rs, off + ((c16 ^ (row & 7)) << 4), 0, 0);
            }
        }
    }
#endif
}
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE opt-in: cached per (kernel, device) -- a process with handles on two
// GPUs must set it on both (a process-wide flag left the second device without it, and its large-LDS launches failed
// instead of falling back to the direct kernel).
template <typename K>
static bool lds_opt_in(K kernel, int bytes) {
    static bool done[64] = {}, ok[64] = {};  // (one pair of tables per kernel instantiation)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!done[dev]) {
        ok[dev] = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
        done[dev] = true;
    }
    return ok[dev];
}
template <int N1, int N2>
static bool launch_pipe_mix(const AttnP& p, hipStream_t st) {
    const int n_items = p.batch * p.n_head;
    const int grid = n_items < 256 ? n_items : 256;
    constexpr int LDSB = 2 * ((N2 + 3) / 4) * 1024 + 4 * (2 * APIPE_IMG + 1024);
    static_assert(LDSB <= 160 * 1024, "LDS");
    if (!lds_opt_in(attn_bf16_pipe_mix_kernel<N1, N2>, LDSB)) return false;
    hipLaunchKernelGGL((attn_bf16_pipe_mix_kernel<N1, N2>), dim3(grid), dim3(192), LDSB, st, p, n_items);
    return true;
}
template <int N1>
static bool launch_pipe_dec(const AttnP& p, hipStream_t st) {
    const int n_items = p.batch * p.n_head;
    const int grid = n_items < 256 ? n_items : 256;  // one workgroup per CU (4 x 26 KB of K|V buffers); a multiple of the 4 heads
    if (!lds_opt_in(attn_bf16_pipe_dec_kernel<N1>, 8 * APIPE_IMG)) return false;
    hipLaunchKernelGGL((attn_bf16_pipe_dec_kernel<N1>), dim3(grid), dim3(192), 8 * APIPE_IMG, st, p, n_items);
    return true;
}
template <int N1, int SH>
static bool launch_pipe(const AttnP& p, hipStream_t st) {
    const int n_items = p.batch * p.n_head;
    int cap = 512;  // two workgroups per CU; a multiple of the head count (a workgroup's items share the head)
#ifdef M3PC_LAB
    static const int env_cap = M3PC_ENV("M3PC_ATTN_PIPE_GRID") ? atoi(M3PC_ENV("M3PC_ATTN_PIPE_GRID")) : 0;
    if (env_cap > 0) cap = env_cap;
#endif
    cap -= cap % p.n_head;
    const int grid = n_items < cap ? n_items : cap;
    if (!lds_opt_in(attn_bf16_pipe_kernel<N1, SH>, 2 * APIPE_BUF)) return false;  // (78 KB of dynamic LDS refused: the direct kernel takes the launch)
    size_t lds_bytes = 2 * APIPE_BUF;
#ifdef M3PC_LAB  // (occupancy experiment: a smaller allocation than the kernel uses -- out-of-range LDS accesses are dropped -- timing only)
    static const int env_lds = M3PC_ENV("M3PC_ATTN_PIPE_LDS") ? atoi(M3PC_ENV("M3PC_ATTN_PIPE_LDS")) : 0;
    if (env_lds > 0) lds_bytes = env_lds;
#endif
    hipLaunchKernelGGL((attn_bf16_pipe_kernel<N1, SH>), dim3(grid), dim3(192), lds_bytes, st, p, n_items);
    return true;
}
static bool wide_off() {  // A/B switch (lab build): the T = 64 shapes on the direct kernel
    static const bool off = M3PC_ENV("M3PC_NO_ATTN_PIPE_WIDE") != nullptr;
    return off;
}
template <int NQ1, int SHQ, int NK1, int SHK, bool PRE>
static bool launch_pipe_wide(const AttnP& p, hipStream_t st) {
    const int n_items = p.batch * p.n_head;
    const int grid = n_items < 256 ? n_items : 256;  // one workgroup per CU (150 KB of row images); a multiple of the 4 heads
    if (!lds_opt_in(attn_bf16_pipe_wide_kernel<NQ1, SHQ, NK1, SHK, PRE>, WPIPE_LDS)) return false;
    constexpr int NTHR = ((NQ1 + SHQ + 31) / 32 + 1) * 64;
    hipLaunchKernelGGL((attn_bf16_pipe_wide_kernel<NQ1, SHQ, NK1, SHK, PRE>), dim3(grid), dim3(NTHR), WPIPE_LDS, st, p, n_items);
    return true;
}
// the shapes the pipelined kernels are built for (the encoder layers and the decoder of the T = 32 and T = 64 candidate passes); everything
// else takes the kernels below
static bool try_pipe(const AttnP& p, hipStream_t st) {
    if (p.hd != 128 || p.n_head != 4 || p.no_pipe == 1 || p.batch * p.n_head < 1024) return false;
    if (p.pre_m) {  // the decoder of an rtg_guiding candidate pass: 32 batch-shared queries, the candidate's own 49 K|V rows, pre-reduced block
        if (p.q_bstride != 0 || p.Q2 || p.K2 || p.Lq < 1 || p.L2 != 0 || !p.pre_l || !p.pre_O) return false;
        if (((uintptr_t)p.Q | (uintptr_t)p.K1 | (uintptr_t)p.V1 | (uintptr_t)p.O | (uintptr_t)p.pre_O) & 15) return false;
        if ((p.ldq | p.ldkv1 | p.ldo) % 8 || (p.kv1_bstride | p.o_bstride) % 8) return false;
        if ((long long)128 * p.ldkv1 * 2 >= 0x7fffffffLL || (long long)128 * p.ldo * 2 >= 0x7fffffffLL) return false;
        if (p.Lq <= 32 && p.L1 == 49) return launch_pipe_dec<49>(p, st);
        if (p.Lq > 32 && p.Lq <= 64 && p.L1 == 97 && !wide_off()) return launch_pipe_wide<0, 64, 97, 0, true>(p, st);  // T = 64
        return false;
    }
    if (((uintptr_t)p.Q | (uintptr_t)p.K1 | (uintptr_t)p.V1 | (uintptr_t)p.O) & 15) return false;
    if ((p.ldq | p.ldkv1 | p.ldo) % 8 || (p.q_bstride | p.kv1_bstride | p.o_bstride) % 8) return false;
    // the decoder of a critic_lambda_guiding candidate pass: 1..4 own queries + shared ones in one tile, own 49 + shared 79 keys
    if (p.Q2 && p.K2 && p.V2 && p.Lq >= 1 && p.Lq <= 4 && p.Lq + p.Lq2 <= 32 && p.L1 == 49 && p.L2 == 79 && p.orow1 == 0 && p.orow2 == p.Lq &&
        !(((uintptr_t)p.Q2 | (uintptr_t)p.K2 | (uintptr_t)p.V2) & 15) && (p.ldq2 | p.ldkv2) % 8 == 0 && (long long)128 * p.ldkv2 * 2 < 0x7fffffffLL &&
        (long long)64 * p.ldkv1 * 2 < 0x7fffffffLL && (long long)64 * p.ldo * 2 < 0x7fffffffLL)
        return launch_pipe_mix<49, 79>(p, st);
    if ((long long)64 * p.ldq * 2 >= 0x7fffffffLL || (long long)64 * p.ldkv1 * 2 >= 0x7fffffffLL || (long long)64 * p.ldo * 2 >= 0x7fffffffLL) return false;
    if (!p.Q2 && !p.K2 && p.Lq == 49 && p.L1 == 49 && p.L2 == 0 && p.orow1 >= 0) {
        return launch_pipe<49, 0>(p, st);
    }
    // T = 64 (BASELINE config 4): 97 rows per candidate; first layer 33 own + 64 shared
    if ((long long)128 * p.ldq * 2 >= 0x7fffffffLL || (long long)128 * p.ldkv1 * 2 >= 0x7fffffffLL || (long long)128 * p.ldo * 2 >= 0x7fffffffLL) return false;
    if (!p.Q2 && !p.K2 && p.Lq == 97 && p.L1 == 97 && p.L2 == 0 && p.orow1 >= 0 && !wide_off()) return launch_pipe_wide<97, 0, 97, 0, false>(p, st);
    if (p.Q2 && p.K2 && p.V2 && p.Lq == 33 && p.L1 == 33 && p.Lq2 == 64 && p.L2 == 64 && p.orow1 == 64 && p.orow2 == 0 && !wide_off() &&
        !(((uintptr_t)p.Q2 | (uintptr_t)p.K2 | (uintptr_t)p.V2) & 15) && (p.ldq2 | p.ldkv2) % 8 == 0 && (long long)128 * p.ldkv2 * 2 < 0x7fffffffLL) {
        return launch_pipe_wide<33, 64, 33, 64, false>(p, st);
    }
    // first layer: the history tokens' rows are shared by the batch (run_block: 17 own + 32 shared rows, shared rows first in the output)
    if (p.Q2 && p.K2 && p.V2 && p.Lq == 17 && p.L1 == 17 && p.Lq2 == 32 && p.L2 == 32 && p.orow1 == 32 && p.orow2 == 0 &&
        !(((uintptr_t)p.Q2 | (uintptr_t)p.K2 | (uintptr_t)p.V2) & 15) && (p.ldq2 | p.ldkv2) % 8 == 0 && (long long)64 * p.ldkv2 * 2 < 0x7fffffffLL) {
        return launch_pipe<17, 32>(p, st);
    }
    return false;
}

// Windows of at most 16 tokens (the zero-shot passes of BASELINE config 5: 10 and 12 kept tokens): the direct kernel gives every
// (window, head) a 32-query x 64-key tile that is 14 % full.  Here TWO windows share one tile -- queries and keys of window w
// at slots 16 w .. 16 w + 15 -- behind a block-diagonal mask.  Bit-identical to the direct kernel: a score is a sum over the
// head dimension only; a window's keys fill positions 0.. of their OWN 16-key step of P V exactly as they fill step 0 there, the
// other window's step multiplies exact zeros; and the softmax's register order meets the same values in the same order,
// preceded by exact -inf / 0 (the 16-slot shift is 8 accumulator registers).  One query / key segment, no pre-reduced block.
template <int HDT>
__global__ __launch_bounds__(64, 4) void attn_bf16_pack2_kernel(AttnP p) {
    constexpr int HD = HDT * 32;
    constexpr int ROWB = HD * 2 + 16;
    constexpr int NS = HD / 16;
    constexpr int CPR = HD / 8;
    __shared__ __attribute__((aligned(16))) char lds[32 * ROWB];
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b0 = 2 * blockIdx.x;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // slot l31: window b0 + (l31 >> 4), its token l31 & 15 (query row and key row alike)
    const int wb = b0 + (l31 >> 4), tk = l31 & 15;
    const bool wok = wb < p.batch;
    const bf16_t* qrow = wok && tk < p.Lq ? (const bf16_t*)p.Q + (long long)wb * p.q_bstride + head * HD + (long long)tk * p.ldq : nullptr;
    const bf16_t* krow = wok && tk < p.L1 ? (const bf16_t*)p.K1 + (long long)wb * p.kv1_bstride + head * HD + (long long)tk * p.ldkv1 : nullptr;
    u32x4 qf[NS], kf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        qf[s] = zero4;
        if (qrow) qf[s] = *(const u32x4*)(qrow + 16 * s + 8 * lh);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        kf[s] = zero4;
        if (krow) kf[s] = *(const u32x4*)(krow + 16 * s + 8 * lh);
    }
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[s]), __builtin_bit_cast(bf16x8, qf[s]), sacc, 0, 0, 0);
    // V image: row r = slot r (zero rows for tokens that do not exist)
    constexpr int VPT = 32 * CPR / 64;
    u32x4 vv[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int x = lane + i * 64, r = x / CPR, kc = x % CPR;
        const int vb = b0 + (r >> 4), vt = r & 15;
        vv[i] = zero4;
        if (vb < p.batch && vt < p.L1) vv[i] = *(const u32x4*)((const bf16_t*)p.V1 + (long long)vb * p.kv1_bstride + head * HD + (long long)vt * p.ldkv1 + kc * 8);
    }
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j = (e & 3) + 8 * (e >> 2) + 4 * lh;  // key slot
        const float v = ((j >> 4) == (l31 >> 4) && (j & 15) < p.L1) ? sacc[e] * p.scale : -INFINITY;
        sacc[e] = v;
        m = fmaxf(m, v);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = __builtin_amdgcn_exp2f((sacc[e] - m) * 1.44269504088896340736f);
        sacc[e] = v;
        l += v;
    }
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int x = lane + i * 64, r = x / CPR, kc = x % CPR;
        *(u32x4*)(lds + r * ROWB + kc * 16) = vv[i];
    }
    __syncthreads();
    f32x16 oacc[HDT];
#pragma unroll
    for (int d = 0; d < HDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    const int gi = lane & 15;
    const int tr_off = (gi >> 2) * ROWB + (((lane >> 4) & 1) * 16 + (gi & 3) * 4) * 2;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pa;
#pragma unroll
        for (int e = 0; e < 8; ++e) pa[e] = (bf16_t)(sacc[8 * s2 + e] * inv);
        const int kb = 16 * s2 + 4 * lh;
#pragma unroll
        for (int d = 0; d < HDT; ++d) {
            const char* base = lds + kb * ROWB + d * 64 + tr_off;
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 8 * ROWB));
            const s16x8 vb = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, __builtin_bit_cast(bf16x8, vb), oacc[d], 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * lh;  // query slot
        const int ob = b0 + (i >> 4), ot = i & 15;
        if (ob < p.batch && ot < p.Lq) {
            bf16_t* Ob = (bf16_t*)p.O + (long long)ob * p.o_bstride + head * HD + (long long)(p.orow1 + ot) * p.ldo;
#pragma unroll
            for (int d = 0; d < HDT; ++d) Ob[d * 32 + l31] = (bf16_t)oacc[d][e];
        }
    }
}

template <int HDT, int NW, int NKT>
static void launch_direct(const AttnP& p, int slots, hipStream_t st) {
    const size_t smem = (size_t)NKT * 32 * (HDT * 64 + 16) + NW * 32 * sizeof(float);
    const int qgroups = (slots + NW * 32 - 1) / (NW * 32);
    hipLaunchKernelGGL((attn_bf16_direct_kernel<HDT, NW, NKT>), dim3(p.batch, p.n_head, qgroups), dim3(NW * 64), smem, st, p);
}

template <int HDT, int NCH>
static void launch_nch(const AttnP& p, dim3 grid, dim3 block, size_t smem, hipStream_t st) {
    hipLaunchKernelGGL((attn_bf16_kernel<HDT, NCH>), grid, block, smem, st, p);
}

template <int HDT>
static void launch_hd(const AttnP& p, hipStream_t st) {
    const int Lk = p.L1 + p.L2;
    const int slots = p.Q2 ? ((p.Lq + 31) & ~31) + p.Lq2 : p.Lq;  // query slots (see AttnP::Q2)
    const int qgroups = (slots + 127) / 128;
    const int nw = slots > 64 ? 4 : 2;  // at least 2 waves so staging has 128 lanes
    const int rows = nw * 32 > 64 ? nw * 32 : 64;
    const size_t smem = (size_t)rows * (HDT * 64 + 16) + 512;  // + per-query weights of the pre-reduced block
    dim3 grid(p.batch, p.n_head, qgroups), block(nw * 64);
    static const bool no_direct = M3PC_ENV("M3PC_NO_ATTN_DIRECT") != nullptr;  // A/B switch
    static const bool no_pipe = M3PC_ENV("M3PC_NO_ATTN_PIPE") != nullptr;      // A/B switch
    if (HDT == 4 && !no_pipe && !no_direct && try_pipe(p, st)) return;
    static const bool no_pack = M3PC_ENV("M3PC_NO_ATTN_PACK") != nullptr;  // A/B switch
    if (!no_direct && !no_pack && p.no_pipe != 1 && !p.Q2 && !p.K2 && !p.pre_m && p.L2 == 0 && p.Lq >= 1 && p.Lq <= 16 && p.L1 >= 1 && p.L1 <= 16 &&
        p.batch >= 64) {  // two windows per tile (the zero-shot passes)
        hipLaunchKernelGGL((attn_bf16_pack2_kernel<HDT>), dim3((p.batch + 1) / 2, p.n_head), dim3(64), 0, st, p);
        return;
    }
    if (Lk <= 64 && !no_direct) {
        if (slots <= 32)
            launch_direct<HDT, 1, 2>(p, slots, st);
        else
            launch_direct<HDT, 2, 2>(p, slots, st);
    } else if (Lk <= 128 && !no_direct) {  // (two waves per block: the 128-row V image divides over 128 threads)
        launch_direct<HDT, 2, 4>(p, slots, st);
    } else if (Lk <= 64)
        launch_nch<HDT, 1>(p, grid, block, smem, st);
    else if (Lk <= 128)
        launch_nch<HDT, 2>(p, grid, block, smem, st);
    else
        launch_nch<HDT, 4>(p, grid, block, smem, st);
}

// one wave per (head, query): scores against the L2 shared keys, their max / exp-sum and the exp-weighted V sum.
// Runs once per (weights, mask, mode) when the decoder tables are built, so it is written for clarity, not speed.
__global__ __launch_bounds__(64) void attn_prestats_kernel(AttnP p, float* pre_m, float* pre_l, float* pre_O) {
    const int head = blockIdx.x, qi = blockIdx.y, lane = threadIdx.x;
    const int HD = p.hd;
    __shared__ float sc[256];
    const bf16_t* q = (const bf16_t*)p.Q + (long long)qi * p.ldq + head * HD;
    const bf16_t* K = (const bf16_t*)p.K2 + head * HD;
    const bf16_t* V = (const bf16_t*)p.V2 + head * HD;
    float m = -INFINITY;
    for (int j = 0; j < p.L2; ++j) {
        float s = 0.f;
        for (int c = lane; c < HD; c += 64) s = fmaf((float)q[c], (float)K[(long long)j * p.ldkv2 + c], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        s *= p.scale;
        if (lane == 0) sc[j] = s;
        m = fmaxf(m, s);
    }
    __syncthreads();
    float l = 0.f;
    for (int j = 0; j < p.L2; ++j) l += __builtin_amdgcn_exp2f((sc[j] - m) * 1.44269504088896340736f);
    for (int c = lane; c < HD; c += 64) {
        float o = 0.f;
        for (int j = 0; j < p.L2; ++j)
            o = fmaf(__builtin_amdgcn_exp2f((sc[j] - m) * 1.44269504088896340736f), (float)V[(long long)j * p.ldkv2 + c], o);
        pre_O[((long long)head * p.Lq + qi) * HD + c] = o;
    }
    if (lane == 0) {
        pre_m[head * p.Lq + qi] = m;
        pre_l[head * p.Lq + qi] = l;
    }
}
void launch_attention_prestats(const AttnP& p, float* pre_m, float* pre_l, float* pre_O, hipStream_t st) {
    hipLaunchKernelGGL(attn_prestats_kernel, dim3(p.n_head, p.Lq), dim3(64), 0, st, p, pre_m, pre_l, pre_O);
}

void launch_attention_bf16(const AttnP& p, hipStream_t st) {
#ifdef M3PC_LAB
    static const bool log_shapes = M3PC_ENV("M3PC_ATTN_LOG") != nullptr;  // (which shapes a workload launches)
    if (log_shapes)
        fprintf(stderr, "attn bf16: batch %d heads %d hd %d  Lq %d Lq2 %d  L1 %d L2 %d  pre %d  orow %d %d  q_bstride %lld ldq %d ldkv1 %d\n", p.batch,
                p.n_head, p.hd, p.Lq, p.Lq2, p.L1, p.L2, p.pre_m != nullptr, p.orow1, p.orow2, p.q_bstride, p.ldq, p.ldkv1);
#endif
    switch (p.hd) {
        case 32: launch_hd<1>(p, st); break;
        case 64: launch_hd<2>(p, st); break;
        case 128: launch_hd<4>(p, st); break;
        default: break;  // validated by the caller
    }
}

}  // namespace m3pc
