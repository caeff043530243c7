// fp32 MFMA GEMM for the FEW-ROW passes (batch-1 policy pass, top-k re-score; M <= ~1k rows).
//
// With so few rows there are not enough 64x64 tiles to fill 1024 SIMDs, and one tile's K loop is a chain of
// dependent v_mfma_f32_32x32x2_f32 (64 clocks each): gemm.hip's kernel therefore splits K over workgroups into
// global slabs and needs a second (reduce) launch.  Here the K split lives INSIDE a 16-wave workgroup:
//   * wave (q, ks) owns 32x32 sub-tile q of the workgroup's tile and K slice ks (K/KS values): a chain of only
//     K/KS/2 MFMAs;
//   * operands go global -> registers directly in MFMA fragment order (lane (row, h) reads the 16 bytes holding
//     k = 8s+4h .. +3 of its row, the same k permutation as gemm.hip, so every product is an exact fp32 fma); no
//     LDS staging, no barrier in the K loop, two 4-step batches of loads in flight per wave;
//   * the KS partial tiles meet in LDS and are summed in slice order (deterministic), then bias / row-table /
//     exact-erf GELU / residual run once per output element.
// One launch per GEMM instead of two, K chains 4-16x shorter.  Costs twice the L2->CU operand traffic of an
// LDS-shared tile, which does not matter at these sizes (<= 200 MB per launch).
#include <cstdlib>

#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int map_row_d(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}
__device__ __forceinline__ float gelu_exact_d(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum_d(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// TQ = 32x32 sub-tiles per workgroup tile (4: 64x64 tile, 1: 32x32 tile), KS = K slices; TQ * KS = 16 waves
template <int TQ, int KS>
__device__ __forceinline__ void gemm_f32_direct_body(const GemmP& p, int bid, float* part, float (*lnst)[2]) {
    static_assert(TQ * KS == 16, "16 waves per workgroup");
    constexpr int BT = TQ == 4 ? 64 : 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = wid % TQ, ks = wid / TQ;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ntn = p.N / BT;
    const int tm = bid / ntn, tn = bid % ntn;
    const int rq = TQ == 4 ? 32 * (q >> 1) : 0, cq = TQ == 4 ? 32 * (q & 1) : 0;
    const int kslice = p.K / KS;
    int gr = tm * BT + rq + l31;
    if (gr >= p.M) gr = p.M - 1;
    const float* a = (const float*)p.A + (long long)map_row_d(p.amap, gr) * p.lda + ks * kslice + 4 * lh;
    const float* w = (const float*)p.W + (long long)(tn * BT + cq + l31) * p.ldw + ks * kslice + 4 * lh;

    // optional LayerNorm of the A rows (TQ == 1): wave w computes the statistics of tile rows 2w, 2w+1 exactly as
    // layernorm_vec_kernel does (same loads per lane, same summation order), the K loop normalises what it loads
    const bool a_ln = TQ == 1 && p.a_ln_g != nullptr;
    float mean = 0.f, rstd = 1.f;
    if (a_ln) {
        const int nsl = p.K >> 8;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            int r = tm * BT + 2 * wid + rr;
            if (r >= p.M) r = p.M - 1;
            const float* x = (const float*)p.A + (long long)map_row_d(p.amap, r) * p.lda;
            f32x4 v[8];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < nsl) {
                    v[i] = *(const f32x4*)(x + i * 256 + lane * 4);
                    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                }
            const float inv_d = 1.0f / (float)p.K;
            const float m = wave_sum_d(s) * inv_d;
            float qq = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < nsl) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = v[i][e] - m;
                        qq += c * c;
                    }
                }
            const float rs = rsqrtf(wave_sum_d(qq) * inv_d + 1e-5f);
            if (lane == 0) {
                lnst[2 * wid + rr][0] = m;
                lnst[2 * wid + rr][1] = rs;
            }
        }
        __syncthreads();
        mean = lnst[l31][0];
        rstd = lnst[l31][1];
    }
    const float* lg = a_ln ? p.a_ln_g + ks * kslice + 4 * lh : nullptr;
    const float* lb = a_ln ? p.a_ln_b + ks * kslice + 4 * lh : nullptr;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 fa0[4], fw0[4], fa1[4], fw1[4];
    const int nb = kslice / 32;  // batches of 4 s-steps (32 k values)
    auto load = [&](f32x4* fa, f32x4* fw, int b) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fa[s] = *(const f32x4*)(a + b * 32 + 8 * s);
            fw[s] = *(const f32x4*)(w + b * 32 + 8 * s);
            if (a_ln) {
                const f32x4 g = *(const f32x4*)(lg + b * 32 + 8 * s);
                const f32x4 bb = *(const f32x4*)(lb + b * 32 + 8 * s);
#pragma unroll
                for (int e = 0; e < 4; ++e) fa[s][e] = (fa[s][e] - mean) * rstd * g[e] + bb[e];
            }
        }
    };
    auto mul = [&](const f32x4* fa, const f32x4* fw) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][e], fw[s][e], acc, 0, 0, 0);
    };
    load(fa0, fw0, 0);
    for (int b = 0; b < nb; b += 2) {
        if (b + 1 < nb) load(fa1, fw1, b + 1);
        mul(fa0, fw0);
        if (b + 2 < nb) load(fa0, fw0, b + 2);
        if (b + 1 < nb) mul(fa1, fw1);
    }

    float* mine = part + (ks * TQ + q) * 1024;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) mine[reg * 64 + lane] = acc[reg];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TQ; ++i) {
        const int x = tid + i * 1024, qq = x >> 10, idx = x & 1023, reg = idx >> 6, ln = idx & 63;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) v += part[(k * TQ + qq) * 1024 + idx];
        const int r = tm * BT + (TQ == 4 ? 32 * (qq >> 1) : 0) + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5);
        const int c = tn * BT + (TQ == 4 ? 32 * (qq & 1) : 0) + (ln & 31);
        if (r < p.M) {
            if (p.bias) v += p.bias[c];
            if (p.rowtab) v += p.rowtab[(long long)(r % p.rt_mod) * p.rt_ld + c];
            if (p.gelu) v = gelu_exact_d(v);
            const long long pr = map_row_d(p.cmap, r);
            if (p.res) v += p.res[pr * p.ldr + c];
            p.Cf[pr * p.ldc + c] = v;
        }
    }
}

template <int TQ, int KS>
__global__ __launch_bounds__(1024) void gemm_f32_direct_kernel(GemmP p) {
    __shared__ float part[16 * 1024];  // [ks][q][reg][lane]
    __shared__ float lnst[32][2];
    gemm_f32_direct_body<TQ, KS>(p, blockIdx.x, part, lnst);
}

// up to four independent problems in one launch: blockIdx.y picks the problem (the four decoder-embedding GEMMs of a
// forward, one per modality, are 4 launches of ~7 us each otherwise)
struct GemmGroupP {
    GemmP p[4];
};
__global__ __launch_bounds__(1024) void gemm_f32_direct_group_kernel(GemmGroupP g) {
    __shared__ float part[16 * 1024];
    __shared__ float lnst[32][2];
    const GemmP& p = g.p[blockIdx.y];
    const int tiles = ((p.M + 31) / 32) * (p.N / 32);
    if ((int)blockIdx.x >= tiles) return;
    gemm_f32_direct_body<1, 16>(p, blockIdx.x, part, lnst);
}

// ------------------------------------------------------------------------------------------------------------------
// The scalar output heads of a few-row fp32 pass (the re-score's rewards / returns heads, mtm_model.py:428-433) in ONE launch:
//   y[r] = w2 . gelu(W1 LN_head(LN_dec(x_r)) + b1) + b2   [de-tokenised]
// for up to two heads (blockIdx.y) over `rows` rows each.  Replaces, per head, a double-LayerNorm launch, a 512 x 512 GEMM launch
// and the Linear(512, 1) launch -- six dependent launches of a chain that is bound by its launches.
// The product runs as gemm_f32_direct_body does (32 x 32 tile per 16-wave workgroup, K split 16 ways inside it, operands global ->
// registers in MFMA fragment order); the two LayerNorms ride on the operand load (every workgroup computes the statistics of its 32
// rows: layernorm_vec_kernel's arithmetic, twice); the epilogue multiplies the gelu'd tile by its 32 entries of w2 and sums them per
// row; the N / 32 partial sums of a row meet in `part`, and the LAST workgroup of a row tile (an atomic ticket) adds them in tile
// order -- a fixed order, so the result does not depend on which workgroup came last.
template <int D>
__global__ __launch_bounds__(1024) void head_f32_fused_kernel(HeadFusedP hp) {
    __shared__ float part[16 * 1024];
    __shared__ float lnst[32][4];
    __shared__ int last_sh;
    const int hs = blockIdx.y;
    // (per-head pointers picked with selects: a run-time index into the by-value parameter struct would copy it to scratch)
    const float* const hW1 = hs ? hp.W1[1] : hp.W1[0];
    const float* const hb1 = hs ? hp.b1[1] : hp.b1[0];
    const float* const hw2 = hs ? hp.w2[1] : hp.w2[0];
    const float* const hb2 = hs ? hp.b2[1] : hp.b2[0];
    const float* const hgB = hs ? hp.gB[1] : hp.gB[0];
    const float* const hbB = hs ? hp.bB[1] : hp.bB[0];
    const float* const hmean = hs ? hp.mean[1] : hp.mean[0];
    const float* const hstdv = hs ? hp.stdv[1] : hp.stdv[0];
    float* const hout = hs ? hp.out[1] : hp.out[0];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int K = D, N = D, ntn = N / 32;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    constexpr int kslice = K / 16;
    const int ks = wid;
    auto phys = [&](int r) { return (long long)((r / hp.grp) * hp.row_mod + hs * hp.grp + r % hp.grp); };
    int gr = tm * 32 + l31;
    if (gr >= hp.rows) gr = hp.rows - 1;
    const float* a = hp.X + phys(gr) * hp.ldx + ks * kslice + 4 * lh;
    const float* w = hW1 + (long long)(tn * 32 + l31) * K + ks * kslice + 4 * lh;
    // statistics of the two LayerNorms of tile rows 2 wid, 2 wid + 1 (decoder.norm, then the head's own)
    {
        constexpr int nsl = K >> 8;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            int r = tm * 32 + 2 * wid + rr;
            if (r >= hp.rows) r = hp.rows - 1;
            const float* x = hp.X + phys(r) * hp.ldx;
            f32x4 v[nsl];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < nsl; ++i) {
                    v[i] = *(const f32x4*)(x + i * 256 + lane * 4);
                    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                }
            const float inv_d = 1.0f / (float)K;
            const float m1 = wave_sum_d(s) * inv_d;
            float qq = 0.f;
#pragma unroll
            for (int i = 0; i < nsl; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = v[i][e] - m1;
                        qq += c * c;
                    }
            const float rs1 = rsqrtf(wave_sum_d(qq) * inv_d + 1e-5f);
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < nsl; ++i) {
                    const f32x4 g = *(const f32x4*)(hp.gA + i * 256 + lane * 4), b = *(const f32x4*)(hp.bA + i * 256 + lane * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - m1) * rs1 * g[e] + b[e];
                    s2 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                }
            const float m2 = wave_sum_d(s2) * inv_d;
            float q2 = 0.f;
#pragma unroll
            for (int i = 0; i < nsl; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = v[i][e] - m2;
                        q2 += c * c;
                    }
            const float rs2 = rsqrtf(wave_sum_d(q2) * inv_d + 1e-5f);
            if (lane == 0) {
                lnst[2 * wid + rr][0] = m1;
                lnst[2 * wid + rr][1] = rs1;
                lnst[2 * wid + rr][2] = m2;
                lnst[2 * wid + rr][3] = rs2;
            }
        }
        __syncthreads();
    }
    const float m1 = lnst[l31][0], rs1 = lnst[l31][1], m2 = lnst[l31][2], rs2 = lnst[l31][3];
    const float* gA = hp.gA + ks * kslice + 4 * lh;
    const float* bA = hp.bA + ks * kslice + 4 * lh;
    const float* gB = hgB + ks * kslice + 4 * lh;
    const float* bB = hbB + ks * kslice + 4 * lh;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // (one 8-value step at a time: 16 waves per workgroup leave 128 registers per wave, and a batch of four steps with its four
    // LayerNorm parameter vectors per operand load spilled)
#pragma unroll 2
    for (int o = 0; o < kslice; o += 8) {
        f32x4 fa = *(const f32x4*)(a + o);
        const f32x4 fw = *(const f32x4*)(w + o);
        const f32x4 g1 = *(const f32x4*)(gA + o), b1 = *(const f32x4*)(bA + o), g2 = *(const f32x4*)(gB + o), b2 = *(const f32x4*)(bB + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = (fa[e] - m1) * rs1 * g1[e] + b1[e];
            fa[e] = (y - m2) * rs2 * g2[e] + b2[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fw[e], acc, 0, 0, 0);
    }
    float* mine = part + ks * 1024;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) mine[reg * 64 + lane] = acc[reg];
    __syncthreads();
    {
        const int reg = tid >> 6, ln = tid & 63;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += part[k * 1024 + tid];
        const int r = tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5);
        const int c = tn * 32 + (ln & 31);
        v = gelu_exact_d(v + hb1[c]) * hw2[c];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);  // over the tile's 32 columns (one half wave per row)
        // (agent-scope atomic store: written through to where every XCD sees it -- no fence, hence no L2 write-back, needed)
        if ((ln & 31) == 0 && r < hp.rows) __hip_atomic_store(hp.part + ((long long)hs * hp.rows + r) * ntn + tn, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the last workgroup of this row tile adds the N / 32 partial sums of each of its rows, in tile order.  Ordering without a
    // device-scope fence (an agent-scope release fence writes the XCD's whole L2 back: measured 350 us per launch with
    // __threadfence() here): the partial sums are agent-scope atomic stores, every wave waits for its own to be acknowledged
    // (vmcnt), the workgroup barrier orders them before the ticket, and the reader uses agent-scope atomic loads.
    // What this rests on is the gfx950 ISA, not the HIP memory model: (1) an agent-scope atomic store is emitted as a
    // write-through store with sc1 set -- it bypasses / writes through the issuing XCD's non-coherent L2 to the memory-side
    // cache every XCD shares; (2) vmcnt counts a store down when that write is acknowledged, so after `s_waitcnt vmcnt(0)` +
    // s_barrier every partial sum of this workgroup is visible chip-wide before thread 0's ticket RMW (an L2-bypassing atomic
    // executed at the same memory-side point) is issued; (3) the last workgroup's agent-scope atomic loads (sc1) miss its own
    // L2 by construction.  No other data is handed over.  tests/test_rescore_gpu.py::test_fused_heads_ticket_stress holds the
    // launch to the un-fused chain bit for bit over thousands of launches on two streams; the pass zeroes the tickets again
    // whenever it ends in an error (m3pc_passes.hip:pruned_decoder), so a failed launch cannot leave a count behind.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(hp.ticket + hs * ((hp.rows + 31) / 32) + tm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh = t == ntn - 1;
        if (last_sh) __hip_atomic_store(hp.ticket + hs * ((hp.rows + 31) / 32) + tm, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (ready for the next launch)
    }
    __syncthreads();
    if (!last_sh) return;
    if (tid < 32) {
        const int r = tm * 32 + tid;
        if (r < hp.rows) {
            const float* pr = hp.part + ((long long)hs * hp.rows + r) * ntn;
            float y = 0.f;
            for (int t = 0; t < ntn; ++t) y += __hip_atomic_load(pr + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            y += hb2[0];
            if (hmean) y = __fadd_rn(__fmul_rn(y, hstdv[0]), hmean[0]);  // de-tokenise (continuous.py:86-94)
            hout[r] = y;
        }
    }
}

bool launch_head_f32_fused(const HeadFusedP& p, hipStream_t st) {
    if (p.n_heads < 1 || p.n_heads > 2 || p.rows < 1 || p.rows > 4096 || p.d != 512) return false;
    if (((uintptr_t)p.X & 15) || (p.ldx % 4) || !p.part || !p.ticket || p.grp < 1 || p.row_mod < p.grp * p.n_heads) return false;
    for (int s = 0; s < p.n_heads; ++s)
        if (!p.W1[s] || !p.b1[s] || !p.w2[s] || !p.b2[s] || !p.gB[s] || !p.bB[s] || !p.out[s] || ((uintptr_t)p.W1[s] & 15)) return false;
    hipLaunchKernelGGL((head_f32_fused_kernel<512>), dim3(((p.rows + 31) / 32) * (p.d / 32), p.n_heads), dim3(1024), 0, st, p);
    return true;
}

bool gemm_f32_direct_covers(const GemmP& p) {
    if (!p.Cf || p.Cb || p.M > 1024 || p.M < 1) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 4) || (p.ldw % 4)) return false;
    const long long tiles64 = (long long)((p.M + 63) / 64) * (p.N / 64);
    // Measured (tools/gemm_bench_f32.py, variants 0 / 2): wins where the problem is tiny -- K = 512 with fewer than
    // 200 64x64 tiles (policy pass 6.4 vs 9.6 us, re-score out-proj 11.9 vs 15.7 us); loses to the LDS-shared tile
    // once operands are re-read by many workgroups (M = 784, N >= 1024: 39 vs 25 us) and on K = 2048 (one launch with
    // 128-deep slices: 20.6 us against 7.2 + 7.4 us for split-K slabs + the row-wise reduce, which also applies the
    // LayerNorm that follows).
    // 200: with the bound-driven re-score at its usual 8 candidates (392 rows) the Q|K|V projection (168 tiles) is 3 us faster
    // here, FFN1 (224 tiles) is not: step -15 us against a limit of 128 (M3PC_DIRECT_TILES re-measures).
    static const long long max_tiles = M3PC_ENV("M3PC_DIRECT_TILES") ? atoll(M3PC_ENV("M3PC_DIRECT_TILES")) : 200;
    if (p.K > 512 || tiles64 >= max_tiles) return false;
    if (p.N % 32 != 0 || p.K % (16 * 32) != 0) return false;
    if (p.a_ln_g && (!p.a_ln_b || p.K % 256 != 0 || p.K > 2048 || ((uintptr_t)p.a_ln_g & 15) || ((uintptr_t)p.a_ln_b & 15))) return false;
    return true;
}

// returns false when the problem is not a few-row fp32 GEMM this kernel covers (the caller falls back to gemm.hip)
bool launch_gemm_f32_direct(const GemmP& p, hipStream_t st) {
    if (!gemm_f32_direct_covers(p)) return false;
    const int grid = ((p.M + 31) / 32) * (p.N / 32);
    hipLaunchKernelGGL((gemm_f32_direct_kernel<1, 16>), dim3(grid), dim3(1024), 0, st, p);
    return true;
}

bool launch_gemm_f32_direct_group(const GemmP* ps, int n, hipStream_t st) {
    if (n < 1 || n > 4) return false;
    GemmGroupP g;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (!gemm_f32_direct_covers(ps[i])) return false;
        g.p[i] = ps[i];
        const int t = ((ps[i].M + 31) / 32) * (ps[i].N / 32);
        tiles = t > tiles ? t : tiles;
    }
    for (int i = n; i < 4; ++i) g.p[i] = ps[0];
    hipLaunchKernelGGL(gemm_f32_direct_group_kernel, dim3(tiles, n), dim3(1024), 0, st, g);
    return true;
}

}  // namespace m3pc
