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
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int map_row_d(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}
__device__ __forceinline__ float gelu_exact_d(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// TQ = 32x32 sub-tiles per workgroup tile (4: 64x64 tile, 1: 32x32 tile), KS = K slices; TQ * KS = 16 waves
template <int TQ, int KS>
__global__ __launch_bounds__(1024) void gemm_f32_direct_kernel(GemmP p) {
    static_assert(TQ * KS == 16, "16 waves per workgroup");
    constexpr int BT = TQ == 4 ? 64 : 32;
    __shared__ float part[16 * 1024];  // [ks][q][reg][lane]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = wid % TQ, ks = wid / TQ;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ntn = p.N / BT;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const int rq = TQ == 4 ? 32 * (q >> 1) : 0, cq = TQ == 4 ? 32 * (q & 1) : 0;
    const int kslice = p.K / KS;
    int gr = tm * BT + rq + l31;
    if (gr >= p.M) gr = p.M - 1;
    const float* a = (const float*)p.A + (long long)map_row_d(p.amap, gr) * p.lda + ks * kslice + 4 * lh;
    const float* w = (const float*)p.W + (long long)(tn * BT + cq + l31) * p.ldw + ks * kslice + 4 * lh;

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

// returns false when the problem is not a few-row fp32 GEMM this kernel covers (the caller falls back to gemm.hip)
bool launch_gemm_f32_direct(const GemmP& p, hipStream_t st) {
    if (!p.Cf || p.Cb || p.M > 1024 || p.M < 1) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 4) || (p.ldw % 4)) return false;
    const long long tiles64 = (long long)((p.M + 63) / 64) * (p.N / 64);
    // Measured (tools/gemm_bench_f32.py, variants 0 / 2): wins where the problem is tiny -- K = 512 with fewer than
    // 128 64x64 tiles (policy pass 6.4 vs 9.6 us, re-score out-proj 11.9 vs 15.7 us); loses to the LDS-shared tile
    // once operands are re-read by many workgroups (M = 784, N >= 1024: 39 vs 25 us) and on K = 2048.
    if (p.K > 512 || tiles64 >= 128) return false;
    const bool deep = true;  // 32x32 tiles, 16 K slices (the 64x64 / 4-slice shape is kept for experiments)
    if (deep) {
        if (p.N % 32 != 0 || p.K % (16 * 32) != 0) return false;
        const int grid = ((p.M + 31) / 32) * (p.N / 32);
        hipLaunchKernelGGL((gemm_f32_direct_kernel<1, 16>), dim3(grid), dim3(1024), 0, st, p);
    } else {
        if (p.N % 64 != 0 || p.K % (4 * 32) != 0) return false;
        const int grid = ((p.M + 63) / 64) * (p.N / 64);
        hipLaunchKernelGGL((gemm_f32_direct_kernel<4, 4>), dim3(grid), dim3(1024), 0, st, p);
    }
    return true;
}

}  // namespace m3pc
