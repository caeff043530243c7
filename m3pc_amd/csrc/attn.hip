// softmax(Q K^T / sqrt(hd)) V on the matrix cores, whole key range resident in registers.
//
// Block = (batch element, head, group of up to 4 query tiles); wave w owns 32 queries.
// Scores are computed TRANSPOSED, S^T = K Q^T (A = K rows j, B = Q rows i), so that after the MFMA a lane
// owns one query column i and the key index j runs over its registers (+ the other lane half):
// the softmax reductions are register-local plus one cross-half exchange, and the normalised P^T
// accumulator is directly the A operand of O = P V (rows of P^T are the k index of the second product),
// so P never moves between lanes and never touches LDS.  Output tile O[i][dim] has dim on the lane =>
// coalesced stores.
//
// Sequence lengths here are tiny (L <= 256), so all S^T tiles of a query tile stay in accumulators
// (16 regs per 32 keys) and no online-softmax rescaling is needed.
//
// Arithmetic: operands are staged to LDS as fp32 (bf16 inputs are widened) and multiplied with
// v_mfma_f32_32x32x2_f32; softmax in fp32.
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) {
    return *(const f32x4*)p;
}
template <>
__device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    const uint2 u = *(const uint2*)p;
    f32x4 r;
    r[0] = __builtin_bit_cast(float, u.x << 16);
    r[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, u.y << 16);
    r[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
    return r;
}

// HDT = hd / 32, NCH = number of 64-key chunks kept in registers
template <typename T, int HDT, int NCH>
__global__ __launch_bounds__(256) void attn_kernel(AttnP p) {
    constexpr int HD = HDT * 32;
    constexpr int ROWF = HD + 4;  // padded LDS row (floats): 16-byte pad keeps float4 frag reads conflict-free
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int b = blockIdx.x, head = blockIdx.y, qg = blockIdx.z;
    const int q0 = qg * 128;
    const int Lk = p.L1 + p.L2;
    const int l31 = lane & 31, lh = lane >> 5;

    const T* Qb = (const T*)p.Q + b * p.q_bstride + head * HD;
    const T* K1 = (const T*)p.K1 + b * p.kv1_bstride + head * HD;
    const T* V1 = (const T*)p.V1 + b * p.kv1_bstride + head * HD;
    const T* K2 = p.K2 ? (const T*)p.K2 + head * HD : nullptr;
    const T* V2 = p.V2 ? (const T*)p.V2 + head * HD : nullptr;

    // ---- stage Q (nw*32 rows) and pull this wave's Q fragments into registers
    for (int c = tid; c < nw * 32 * (HD / 4); c += nthr) {
        const int r = c / (HD / 4), k4 = c % (HD / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + r < p.Lq) v = load4<T>(Qb + (long long)(q0 + r) * p.ldq + k4 * 4);
        *(f32x4*)(lds + r * ROWF + k4 * 4) = v;
    }
    __syncthreads();
    f32x4 qf[HDT * 4];  // quarter s: floats k = 8s+4h .. +3 of query row (wid*32 + l31)
#pragma unroll
    for (int s = 0; s < HDT * 4; ++s) qf[s] = *(const f32x4*)(lds + (wid * 32 + l31) * ROWF + 8 * s + 4 * lh);
    __syncthreads();

    f32x16 sacc[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[c][jt][e] = 0.f;

    // ---- S^T = K Q^T
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j0 = c * 64;
        if (j0 < Lk) {
            for (int x = tid; x < 64 * (HD / 4); x += nthr) {
                const int r = x / (HD / 4), k4 = x % (HD / 4), j = j0 + r;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (j < p.L1)
                    v = load4<T>(K1 + (long long)j * p.ldkv1 + k4 * 4);
                else if (j < Lk)
                    v = load4<T>(K2 + (long long)(j - p.L1) * p.ldkv2 + k4 * 4);
                *(f32x4*)(lds + r * ROWF + k4 * 4) = v;
            }
            __syncthreads();
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                if (j0 + jt * 32 < Lk) {
#pragma unroll
                    for (int s = 0; s < HDT * 4; ++s) {
                        const f32x4 kf = *(const f32x4*)(lds + (jt * 32 + l31) * ROWF + 8 * s + 4 * lh);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            sacc[c][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[s][e], sacc[c][jt], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- softmax over j for the lane's query column.  reg e of tile (c,jt) is key
    //      j = c*64 + jt*32 + (e&3) + 8*(e>>2) + 4*lh
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
                const float v = expf(sacc[c][jt][e] - m);
                sacc[c][jt][e] = v;
                l += v;
            }
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[c][jt][e] *= inv;

    // ---- O = P V : A = P (from the S^T accumulator: reg e <-> k = key row), B = V[key][dim]
    f32x16 oacc[HDT];
#pragma unroll
    for (int d = 0; d < HDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j0 = c * 64;
        if (j0 < Lk) {
            for (int x = tid; x < 64 * (HD / 4); x += nthr) {
                const int r = x / (HD / 4), k4 = x % (HD / 4), j = j0 + r;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (j < p.L1)
                    v = load4<T>(V1 + (long long)j * p.ldkv1 + k4 * 4);
                else if (j < Lk)
                    v = load4<T>(V2 + (long long)(j - p.L1) * p.ldkv2 + k4 * 4);
                *(f32x4*)(lds + r * ROWF + k4 * 4) = v;
            }
            __syncthreads();
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                if (j0 + jt * 32 < Lk) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int jr = jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const float pe = sacc[c][jt][e];
#pragma unroll
                        for (int d = 0; d < HDT; ++d) {
                            const float vv = lds[jr * ROWF + d * 32 + l31];
                            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(pe, vv, oacc[d], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- store: oacc[d][e] = O[i = wid*32 + (e&3)+8*(e>>2)+4*lh][dim = d*32 + l31]
    T* Ob = (T*)p.O + b * p.o_bstride + head * HD;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = q0 + wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (i < p.Lq) {
#pragma unroll
            for (int d = 0; d < HDT; ++d) Ob[(long long)i * p.ldo + d * 32 + l31] = (T)oacc[d][e];
        }
    }
}

// Few-sequence variant (batch-1 policy pass, top-k re-score): the kernel above gives one wave a whole query tile and
// walks the keys serially -- 128 dependent f32 MFMAs (64 clocks each) per 32-key tile -- while most of the chip
// idles.  Here a block is ONE 32-query tile of one (batch, head) and its four waves SPLIT THE KEY TILES (wave w takes
// tiles w, w+4; Lk <= 256).  Each wave runs a complete softmax over its own keys (max m_w, sum l_w, un-normalised
// O_w = sum exp(s - m_w) V) out of a wave-private LDS slab; the four partial results are merged the streaming-softmax
// way: m = max m_w, O = sum_w e^{m_w - m} O_w / sum_w e^{m_w - m} l_w.  Same arithmetic as one softmax up to fp32
// rounding of the re-association.
template <typename T, int HDT>
__global__ __launch_bounds__(256) void attn_split_kernel(AttnP p) {
    constexpr int HD = HDT * 32;
    constexpr int ROWF = HD + 4;
    constexpr int SLAB = 32 * ROWF;  // floats: one 32-row operand tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsQ = lds;                    // [32][ROWF]
    float* ldsML = lds + SLAB;            // [4][32][2]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float* slab = lds + SLAB + 256 + wid * SLAB;  // this wave's K / V / O_w tile
    const int b = blockIdx.x, head = blockIdx.y, q0 = blockIdx.z * 32;
    const int Lk = p.L1 + p.L2;
    const int nkt = (Lk + 31) / 32;
    const int l31 = lane & 31, lh = lane >> 5;

    const T* Qb = (const T*)p.Q + b * p.q_bstride + head * HD;
    const T* K1 = (const T*)p.K1 + b * p.kv1_bstride + head * HD;
    const T* V1 = (const T*)p.V1 + b * p.kv1_bstride + head * HD;
    const T* K2 = p.K2 ? (const T*)p.K2 + head * HD : nullptr;
    const T* V2 = p.V2 ? (const T*)p.V2 + head * HD : nullptr;

    for (int c = tid; c < 32 * (HD / 4); c += 256) {
        const int r = c / (HD / 4), k4 = c % (HD / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + r < p.Lq) v = load4<T>(Qb + (long long)(q0 + r) * p.ldq + k4 * 4);
        *(f32x4*)(ldsQ + r * ROWF + k4 * 4) = v;
    }
    __syncthreads();
    f32x4 qf[HDT * 4];
#pragma unroll
    for (int s = 0; s < HDT * 4; ++s) qf[s] = *(const f32x4*)(ldsQ + l31 * ROWF + 8 * s + 4 * lh);

    // K and V never touch LDS here: a lane reads its key row's fragment (16 bytes per k-step) and its V values
    // (one float per (key pair, dim tile)) straight from global memory in MFMA operand order -- there is no reuse
    // inside a wave to stage for, and the loads of a whole tile are in flight together.
    f32x16 sacc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int kt = wid + 4 * t;
        if (kt < nkt) {
            const int j = kt * 32 + l31;
            const T* kr = j < p.L1 ? K1 + (long long)j * p.ldkv1 : (j < Lk ? K2 + (long long)(j - p.L1) * p.ldkv2 : nullptr);
            f32x4 kf[HDT * 4];
#pragma unroll
            for (int s = 0; s < HDT * 4; ++s) {
                kf[s] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (kr) kf[s] = load4<T>(kr + 8 * s + 4 * lh);
            }
#pragma unroll
            for (int s = 0; s < HDT * 4; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s][e], qf[s][e], sacc[t], 0, 0, 0);
        }
    }
    // softmax over this wave's keys for the lane's query column
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = (wid + 4 * t) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float v = (j < Lk) ? sacc[t][e] * p.scale : -INFINITY;
            sacc[t][e] = v;
            m = fmaxf(m, v);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = m == -INFINITY ? 0.f : expf(sacc[t][e] - m);
            sacc[t][e] = v;
            l += v;
        }
    l += __shfl_xor(l, 32);
    if (lh == 0) {
        ldsML[(wid * 32 + l31) * 2 + 0] = m;
        ldsML[(wid * 32 + l31) * 2 + 1] = l;
    }
    // O_w = P V
    f32x16 oacc[HDT];
#pragma unroll
    for (int d = 0; d < HDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int kt = wid + 4 * t;
        if (kt < nkt) {
            float vv[16][HDT];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int jr = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const T* vr = jr < p.L1 ? V1 + (long long)jr * p.ldkv1 : (jr < Lk ? V2 + (long long)(jr - p.L1) * p.ldkv2 : nullptr);
#pragma unroll
                for (int d = 0; d < HDT; ++d) vv[e][d] = vr ? (float)vr[d * 32 + l31] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int d = 0; d < HDT; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[t][e], vv[e][d], oacc[d], 0, 0, 0);
        }
    }
    // park O_w in the wave's slab (row i, dim), then merge: wave w owns dim tile d = w
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
        for (int d = 0; d < HDT; ++d) slab[i * ROWF + d * 32 + l31] = oacc[d][e];
    }
    __syncthreads();
    if (wid < HDT) {
        T* Ob = (T*)p.O + b * p.o_bstride + head * HD;
        const float* slabs = lds + SLAB + 256;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * lh;
            float mw[4], lw[4], mt = -INFINITY;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                mw[w] = ldsML[(w * 32 + i) * 2 + 0];
                lw[w] = ldsML[(w * 32 + i) * 2 + 1];
                mt = fmaxf(mt, mw[w]);
            }
            float lt = 0.f, o = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = mw[w] == -INFINITY ? 0.f : expf(mw[w] - mt);
                lt = fmaf(lw[w], f, lt);
                o = fmaf(slabs[w * SLAB + i * ROWF + wid * 32 + l31], f, o);
            }
            if (q0 + i < p.Lq) Ob[(long long)(q0 + i) * p.ldo + wid * 32 + l31] = (T)(o / lt);
        }
    }
}

// Few sequences AND at most two key tiles (Lk <= 64: the batch-1 policy pass and the top-k re-score at T = 32): with the
// keys split over waves only two of the four waves would work and each would run two 64-MFMA chains.  Here every wave
// works in both phases and every chain is 32 MFMAs:
//   S phase:  wave w takes key tile w & 1 and HALF of the head dimension (w >> 1): Q and K fragments straight from
//             global memory in operand order (no LDS staging, no barrier before the first MFMA); the two partial S^T tiles
//             of a key tile meet in LDS, waves 0 / 1 finish the softmax of their key tile (scale, max, exp, sum);
//   PV phase: P^T (still in accumulator layout = the A operand of P V, one private 64-byte LDS slot per lane) and
//             (max, sum) of both key tiles go through LDS to all four waves; wave w computes output dims 32 w .. +31
//             over both key tiles, with P rescaled by exp(m_kt - m) beforehand, and stores its slice of O directly.
// Two barriers, no O merge.  (fp32: v_mfma_f32_32x32x2_f32, every product an exact fp32 fma.)
template <typename T, int HDT>
__global__ __launch_bounds__(256) void attn_pair_kernel(AttnP p) {
    static_assert(HDT == 4 || HDT == 2, "head dims 128 / 64");
    constexpr int HD = HDT * 32, HH = HDT / 2;  // HH: 32-wide d tiles per half
    __shared__ f32x16 lds_s[2][64];   // partial S^T of the upper d half, per key tile, one slot per lane
    __shared__ f32x16 lds_p[2][64];   // exp(S - m_kt) per key tile, accumulator layout
    __shared__ float lds_ml[2][32][2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int kt = wid & 1, dh = wid >> 1;
    const int b = blockIdx.x, head = blockIdx.y, q0 = blockIdx.z * 32;
    const int Lk = p.L1 + p.L2;

    const T* Qb = (const T*)p.Q + b * p.q_bstride + head * HD;
    const T* K1 = (const T*)p.K1 + b * p.kv1_bstride + head * HD;
    const T* V1 = (const T*)p.V1 + b * p.kv1_bstride + head * HD;
    const T* K2 = p.K2 ? (const T*)p.K2 + head * HD : nullptr;
    const T* V2 = p.V2 ? (const T*)p.V2 + head * HD : nullptr;

    // S^T partial: keys of tile kt x queries, over d in [dh * HD/2, (dh + 1) * HD/2)
    const int jq = q0 + l31, jk = kt * 32 + l31;
    const T* qr = jq < p.Lq ? Qb + (long long)jq * p.ldq + dh * (HD / 2) : nullptr;
    const T* kr = jk < p.L1 ? K1 + (long long)jk * p.ldkv1 + dh * (HD / 2)
                            : (jk < Lk ? K2 + (long long)(jk - p.L1) * p.ldkv2 + dh * (HD / 2) : nullptr);
    f32x4 qf[HH * 4], kf[HH * 4];
#pragma unroll
    for (int s = 0; s < HH * 4; ++s) {
        qf[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        kf[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qr) qf[s] = load4<T>(qr + 8 * s + 4 * lh);
        if (kr) kf[s] = load4<T>(kr + 8 * s + 4 * lh);
    }
    // V values of this wave's output dims for both key tiles (issued now, used after the second barrier)
    float vv[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int jr = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const T* vr = jr < p.L1 ? V1 + (long long)jr * p.ldkv1 : (jr < Lk ? V2 + (long long)(jr - p.L1) * p.ldkv2 : nullptr);
            vv[t][e] = vr ? (float)vr[wid * 32 + l31] : 0.f;
        }
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < HH * 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s][e], qf[s][e], sacc, 0, 0, 0);
    if (dh == 1) lds_s[kt][lane] = sacc;
    __syncthreads();
    if (dh == 0) {
        const f32x16 hi = lds_s[kt][lane];
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float v = (j < Lk) ? (sacc[e] + hi[e]) * p.scale : -INFINITY;
            sacc[e] = v;
            m = fmaxf(m, v);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = m == -INFINITY ? 0.f : expf(sacc[e] - m);
            sacc[e] = v;
            l += v;
        }
        l += __shfl_xor(l, 32);
        lds_p[kt][lane] = sacc;
        if (lh == 0) {
            lds_ml[kt][l31][0] = m;
            lds_ml[kt][l31][1] = l;
        }
    }
    __syncthreads();
    // O[:, 32 wid .. +31] = sum_kt exp(m_kt - m) P_kt V_kt / sum_kt exp(m_kt - m) l_kt   (query = l31 on the A side)
    const float m0 = lds_ml[0][l31][0], l0 = lds_ml[0][l31][1], m1 = lds_ml[1][l31][0], l1 = lds_ml[1][l31][1];
    const float mt = fmaxf(m0, m1);
    const float f0 = m0 == -INFINITY ? 0.f : expf(m0 - mt), f1 = m1 == -INFINITY ? 0.f : expf(m1 - mt);
    f32x16 oacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x16 pt = lds_p[t][lane];
        const float f = t == 0 ? f0 : f1;
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pt[e] * f, vv[t][e], oacc, 0, 0, 0);
    }
    // oacc[e] is (query (e & 3) + 8 (e >> 2) + 4 lh, dim 32 wid + l31): the sums of that QUERY live in lds_ml[.][query]
    T* Ob = (T*)p.O + b * p.o_bstride + head * HD;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float a0 = lds_ml[0][i][0], b0 = lds_ml[0][i][1], a1 = lds_ml[1][i][0], b1 = lds_ml[1][i][1];
        const float am = fmaxf(a0, a1);
        const float g0 = a0 == -INFINITY ? 0.f : expf(a0 - am), g1 = a1 == -INFINITY ? 0.f : expf(a1 - am);
        const float lt = fmaf(b1, g1, b0 * g0);
        if (q0 + i < p.Lq) Ob[(long long)(q0 + i) * p.ldo + wid * 32 + l31] = (T)(oacc[e] / lt);
    }
}

template <typename T, int HDT>
static void launch_split(const AttnP& p, hipStream_t st) {
    constexpr int ROWF = HDT * 32 + 4;
    const size_t smem = (size_t)(5 * 32 * ROWF + 256) * sizeof(float);
    static bool attr_set[64] = {};  // (> 64 KiB of dynamic LDS needs the opt-in once per kernel AND device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)attn_split_kernel<T, HDT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev] = true;
    }
    dim3 grid(p.batch, p.n_head, (p.Lq + 31) / 32), block(256);
    hipLaunchKernelGGL((attn_split_kernel<T, HDT>), grid, block, smem, st, p);
}

template <typename T, int HDT, int NCH>
static void launch_nch(const AttnP& p, dim3 grid, dim3 block, size_t smem, hipStream_t st) {
    static bool attr_set[64] = {};  // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)attn_kernel<T, HDT, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((attn_kernel<T, HDT, NCH>), grid, block, smem, st, p);
}

template <typename T, int HDT>
static void launch_hd(const AttnP& p, hipStream_t st) {
    const int Lk = p.L1 + p.L2;
    const int qgroups = (p.Lq + 127) / 128;
    static const bool no_split = M3PC_ENV("M3PC_NO_ATTN_SPLIT") != nullptr;  // A/B switch
    if (Lk <= 256 && (long long)p.batch * p.n_head * qgroups <= 64 && !no_split) {
        static const bool no_pair = M3PC_ENV("M3PC_NO_ATTN_PAIR") != nullptr;  // A/B switch
        if constexpr (HDT == 4) {  // (each wave owns one 32-wide slice of the output: four waves = head dim 128)
            if (Lk <= 64 && !no_pair) {
                dim3 grid(p.batch, p.n_head, (p.Lq + 31) / 32), block(256);
                hipLaunchKernelGGL((attn_pair_kernel<T, HDT>), grid, block, 0, st, p);
                return;
            }
        }
        launch_split<T, HDT>(p, st);  // few sequences: split the keys over the waves instead
        return;
    }
    // always 4 waves: waves past the last query tile still help staging K/V (their MFMAs run on an otherwise
    // idle SIMD over zero-filled query rows and are never stored)
    const int nw = 4;
    const int rows = nw * 32 > 64 ? nw * 32 : 64;
    const size_t smem = (size_t)rows * (HDT * 32 + 4) * sizeof(float);
    dim3 grid(p.batch, p.n_head, qgroups), block(nw * 64);
    if (Lk <= 64)
        launch_nch<T, HDT, 1>(p, grid, block, smem, st);
    else if (Lk <= 128)
        launch_nch<T, HDT, 2>(p, grid, block, smem, st);
    else
        launch_nch<T, HDT, 4>(p, grid, block, smem, st);
}

template <typename T>
static void launch_t(const AttnP& p, hipStream_t st) {
    switch (p.hd) {
        case 32: launch_hd<T, 1>(p, st); break;
        case 64: launch_hd<T, 2>(p, st); break;
        case 128: launch_hd<T, 4>(p, st); break;
        default: break;  // validated by the caller
    }
}

void launch_attention(const AttnP& p, int dtype, hipStream_t st) {
    if (p.batch <= 0 || p.Lq <= 0) return;
    if (dtype == DT_BF16)
        launch_attention_bf16(p, st);  // native bf16 MFMA kernel (attn_bf16.hip)
    else
        launch_t<float>(p, st);
}

}  // namespace m3pc
