// bf16 MFMA GEMM, 256x256 block tile, 8 waves (2 x 4, wave tile 128 x 64), K streamed through a 4-slot LDS
// ring of 32-deep stages filled by global_load_lds_dwordx4 (LDS-DMA), three stages in flight.
//
// Why this shape (measured on MI355X with tools/gemm_bench.py):
//   * a 128x128 tile needs ~128 B/clk/CU of L2->LDS traffic at full MFMA rate, a 256x256 tile 32 B/clk/CU;
//   * the K loops here are short (K = 512 is 16 stages), so the HBM/L2 latency of every stage must be hidden by
//     stages already in flight, not by a long steady state: bytes in flight per CU = bandwidth x latency
//     ~ 96 KiB, i.e. three 32-KiB stages;
//   * LDS-DMA keeps those bytes out of the VGPRs (accumulators take 128 of the 256 registers a wave gets at
//     two waves per SIMD).
// Synchronisation per stage s (one barrier):  counted s_waitcnt vmcnt leaves the two younger stages in
// flight -> s_barrier (every wave's share of stage s has landed, every wave has finished reading slot
// (s-1)%4) -> issue stage s+3 into slot (s-1)%4 -> multiply stage s.  __syncthreads() is never used in the
// loop: it would drain the LDS-DMA queue.
// LDS rows are 64 B; chunk position c of row r holds logical chunk c ^ ((r >> 2) & 3) (swizzle applied on the
// global SOURCE address, the DMA writes linearly), which makes the 16-row ds_read_b128 groups conflict-free.
#include "gemm_epilogue.h"
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

enum { EPI_GELU = 1, EPI_RES = 2, EPI_ROWTAB = 4, EPI_F32OUT = 8 };

__device__ __forceinline__ float gelu_fast3(float x) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float erf_abs = fmaf(-p, e, 1.0f);
    const float hx = 0.5f * x;
    return fmaf(fabsf(hx), erf_abs, hx);
}
__device__ __forceinline__ int map_row3(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}

#define RING_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")
template <int N>
__device__ __forceinline__ void ring_wait_barrier() {
    static_assert(N == 0 || N == 3 || N == 4 || N == 6 || N == 8, "add the immediate");
    if constexpr (N == 0) RING_WAIT_BARRIER(0);
    if constexpr (N == 3) RING_WAIT_BARRIER(3);
    if constexpr (N == 4) RING_WAIT_BARRIER(4);
    if constexpr (N == 6) RING_WAIT_BARRIER(6);
    if constexpr (N == 8) RING_WAIT_BARRIER(8);
}

// NSLOT = 4: three stages in flight, one block per CU.  NSLOT = 3 (BN = 128 only): two stages in flight, 72 KiB of
// LDS and <= 128 VGPRs so that two blocks share a CU and one block's epilogue overlaps the other's main loop.
template <int BN, int EPI, int NSLOT>
__global__ __launch_bounds__(512, NSLOT == 3 ? 4 : 2) void gemm_ring_kernel(GemmP p) {
    constexpr int BM = 256;
    constexpr int WN = BN / 64, WM = 8 / WN;     // wave grid: 2x4 for BN=256, 4x2 for BN=128
    constexpr int WTM = BM / WM, TM = WTM / 32;  // wave tile WTM x 64
    constexpr int TN = 2;
    constexpr int STAGE = (BM + BN) * 64;        // bytes per ring slot
    constexpr int NI_A = BM / 16 / 8, NI_W = BN / 16 / 8;  // 1-KiB DMA instructions (16 rows) per wave per stage
    __shared__ __attribute__((aligned(1024))) char smem[NSLOT * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    const int ntn = p.N / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * BM, col0 = tn * BN;
    const long long lda_b = (long long)p.lda * 2, ldw_b = (long long)p.ldw * 2;
    const int nst = p.K / 32;

    // DMA descriptors: instruction I = wid + 8 i covers tile rows 16 I .. 16 I + 15 (4 lanes per 64-B row)
    const char* a_src[NI_A];
    const char* w_src[NI_W];
#pragma unroll
    for (int i = 0; i < NI_A; ++i) {
        const int r = 16 * (wid + 8 * i) + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        int gr = row0 + r;
        if (gr >= p.M) gr = p.M - 1;
        a_src[i] = (const char*)p.A + (long long)map_row3(p.amap, gr) * lda_b + q * 16;
    }
#pragma unroll
    for (int i = 0; i < NI_W; ++i) {
        const int r = 16 * (wid + 8 * i) + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        w_src[i] = (const char*)p.W + (long long)(col0 + r) * ldw_b + q * 16;
    }
    const int wave_dst = __builtin_amdgcn_readfirstlane(wid) * 1024;

    const bool dbg_noload = p.variant == 11 || p.variant == 12;   // timing experiments (tools/gemm_bench.py)
    auto issue = [&](int st) {
        if (dbg_noload) return;
        char* base = smem + (st % NSLOT) * STAGE + wave_dst;
        const long long ko = (long long)st * 64;
#pragma unroll
        for (int i = 0; i < NI_A; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + ko), (lptr_t)(base + i * 8192), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NI_W; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + ko), (lptr_t)(base + BM * 64 + i * 8192), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    const int f0 = l31 * 64 + ((lh ^ sw) * 16), f1 = l31 * 64 + (((2 + lh) ^ sw) * 16);
    const int fragA = wr * WTM * 64;
    const int fragW = BM * 64 + wc * 64 * 64;

    auto compute = [&](int st) {
        const char* cur = smem + (st % NSLOT) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fo = s == 0 ? f0 : f1;
            u32x4 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const u32x4*)(cur + fragA + i * 32 * 64 + fo);
#pragma unroll
            for (int j = 0; j < TN; ++j) fw[j] = *(const u32x4*)(cur + fragW + j * 32 * 64 + fo);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fw[j]), acc[i][j], 0, 0, 0);
        }
    };

    // nst >= 4 is guaranteed by the launcher (K >= 128)
    constexpr int PER = NI_A + NI_W;  // DMA instructions per wave per stage
    constexpr int F = NSLOT - 1;      // stages in flight
    static_assert(PER == 4 || PER == 3, "vmcnt immediates assume 3 or 4 DMA instructions per stage");
#pragma unroll
    for (int s = 0; s < F; ++s) issue(s);
    for (int st = 0; st < nst - (F - 1); ++st) {
        ring_wait_barrier<(F - 1) * PER>();
        if (st + F < nst) issue(st + F);
        compute(st);
    }
    // tail: the last F-1 stages (nothing left to issue)
    if constexpr (F == 3) {
        ring_wait_barrier<PER>();
        compute(nst - 2);
    }
    ring_wait_barrier<0>();
    compute(nst - 1);

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    {
        const int wu = __builtin_amdgcn_readfirstlane(wid);
        gemm_epilogue<EPI, TM, TN>(p, acc, row0 + (wu / WN) * WTM, col0 + (wu % WN) * 64, row0, BM, lane);
    }
}

template <int BN, int EPI, int NSLOT>
static void launch_cfg(const GemmP& p, hipStream_t st) {
    const int grid = ((p.M + 255) / 256) * (p.N / BN);
    hipLaunchKernelGGL((gemm_ring_kernel<BN, EPI, NSLOT>), dim3(grid), dim3(512), 0, st, p);
}

template <int BN, int NSLOT>
static bool launch_bn(const GemmP& p, hipStream_t st) {
    const bool f32out = p.Cf != nullptr;
    const int epi = (p.gelu ? EPI_GELU : 0) | (p.res ? EPI_RES : 0) | (p.rowtab ? EPI_ROWTAB : 0) | (f32out ? EPI_F32OUT : 0);
    switch (epi) {
        case 0: launch_cfg<BN, 0, NSLOT>(p, st); return true;
        case EPI_F32OUT: launch_cfg<BN, EPI_F32OUT, NSLOT>(p, st); return true;
        case EPI_GELU: launch_cfg<BN, EPI_GELU, NSLOT>(p, st); return true;
        case EPI_GELU | EPI_F32OUT: launch_cfg<BN, EPI_GELU | EPI_F32OUT, NSLOT>(p, st); return true;
        case EPI_RES | EPI_F32OUT: launch_cfg<BN, EPI_RES | EPI_F32OUT, NSLOT>(p, st); return true;
        case EPI_ROWTAB | EPI_F32OUT: launch_cfg<BN, EPI_ROWTAB | EPI_F32OUT, NSLOT>(p, st); return true;
        default: return false;
    }
}

// returns false when the shape / epilogue is not covered (the caller falls back to gemm.hip's kernel)
bool launch_gemm_ring(const GemmP& p, hipStream_t st) {
    if (p.K % 32 != 0 || p.K < 128 || p.N % 128 != 0) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 8) || (p.ldw % 8)) return false;
    const long long rows256 = (p.M + 255) / 256;
    if (rows256 * (p.N / 128) < 256) return false;  // too few tiles to fill the chip
    if (p.variant == 13) return launch_bn<128, 3>(p, st);
    if (p.variant == 8 || p.N % 256 != 0) return launch_bn<128, 4>(p, st);  // variants 10-12: timing experiments
    return launch_bn<256, 4>(p, st);
}

}  // namespace m3pc
