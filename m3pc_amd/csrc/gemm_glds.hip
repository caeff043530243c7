// bf16 MFMA GEMM with direct-to-LDS staging (global_load_lds_dwordx4) for the many-row GEMMs of the
// candidate pass.  Same math and epilogue as gemm.hip; what changes is how tiles reach LDS:
//   * no VGPR round trip and no ds_write: each wave-instruction drops 8 rows x 128 B (1 KiB) of a tile
//     straight into LDS, lane l -> bytes [16 l, 16 l + 16) of that KiB;
//   * LDS rows are therefore unpadded (128 B); bank conflicts are avoided by an XOR swizzle applied on the
//     SOURCE side: LDS chunk position c of row r holds logical 16-byte chunk c ^ ((r >> 1) & 7), and the
//     fragment reads apply the same involution.  Rows r, r+1 share a 256-byte bank row (different halves),
//     rows two apart get different chunk slots => ds_read_b128 of 16 distinct rows is conflict-free;
//   * two LDS buffers, tile k+1 is in flight while tile k is multiplied, one barrier per k-tile;
//   * residual rows (out-proj / FFN2 epilogues) are prefetched into registers before the K loop, so the
//     epilogue does not start with a dependent HBM round trip.
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { EPI_GELU = 1, EPI_RES = 2, EPI_ROWTAB = 4, EPI_F32OUT = 8 };

__device__ long long g_clock_probe[2];  // {shader clocks, 100-MHz wall ticks} of one workgroup (debug variants only)
void read_clock_probe(long long out[2]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock_probe), 2 * sizeof(long long)); }

__device__ __forceinline__ float gelu_fast2(float x) {
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

__device__ __forceinline__ int map_row2(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

// ROWB = bytes of K per LDS row and stage: 128 (64 k, 4 MFMA k-steps per barrier) or 64 (32 k, 2 k-steps, half the
// LDS => more workgroups per CU, i.e. more tiles in flight against the DMA latency)
//
// One output tile.  `bid` is the tile's index among the `ntm` x (N / BN) tiles that cover rows [row_begin, ...).
// NSLOT = 2: double buffer, one stage in flight, __syncthreads per stage.  NSLOT = 3 (with ROWB = 64): ring of three
// 32-deep stages, two in flight, counted s_waitcnt vmcnt + raw s_barrier (a __syncthreads would drain the DMA queue),
// 48 KiB of LDS per 128x128 workgroup => three workgroups per CU.
template <int N>
__device__ __forceinline__ void glds_wait_barrier() {
    static_assert(N == 0 || N == 2 || N == 4 || N == 6 || N == 8, "add the immediate");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
}
template <int BM, int BN, int EPI, int WM, int WN, int ROWB, int NSLOT = 2>
__device__ __forceinline__ void gemm_glds_tile(const GemmP& p, char* smem, int bid, const int ntm, const int row_begin) {
    constexpr int RPI = 1024 / ROWB;   // tile rows per 1-KiB DMA instruction
    constexpr int LPR = ROWB / 16;     // lanes (16-byte chunks) per row
    constexpr int KS = ROWB / 32;      // MFMA k-steps per stage
    constexpr int NW = WM * WN;                      // waves per block, WM x WN grid of wave tiles
    constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;      // 32x32 MFMA tiles per wave
    constexpr int NI_A = BM / RPI / NW, NI_W = BN / RPI / NW;  // 1-KiB wave-instructions per wave per k-tile
    constexpr int BUF = (BM + BN) * ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    const int ntn = p.N / BN;
    const int nwg = ntm * ntn;
    {
        const int q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = row_begin + tm * BM, col0 = tn * BN;
    const long long lda_b = (long long)p.lda * 2, ldw_b = (long long)p.ldw * 2;
    const int nkt = p.K * 2 / ROWB;

    // staging descriptors: wave-instruction I = wid + 4 i covers tile rows 8 I .. 8 I + 7
    const char* a_src[NI_A];
    const char* w_src[NI_W];
#pragma unroll
    for (int i = 0; i < NI_A; ++i) {
        const int r = RPI * (wid + NW * i) + lane / LPR;
        const int q = (lane % LPR) ^ (ROWB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3));
        int gr = row0 + r;
        if (gr >= p.M) gr = p.M - 1;
        a_src[i] = (const char*)p.A + (long long)map_row2(p.amap, gr) * lda_b + q * 16;
    }
#pragma unroll
    for (int i = 0; i < NI_W; ++i) {
        const int r = RPI * (wid + NW * i) + lane / LPR;
        const int q = (lane % LPR) ^ (ROWB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3));
        w_src[i] = (const char*)p.W + (long long)(col0 + r) * ldw_b + q * 16;
    }
    const int wave_dst = __builtin_amdgcn_readfirstlane(wid) * 1024;

    // timing experiments (tools/gemm_bench.py): 20-22 / 24-25 on the double buffer, 28-30 on the ring
    const bool dbg_noload = p.variant == 20 || p.variant == 22 || p.variant == 28 || p.variant == 30;
    const bool dbg_nostore = p.variant == 21 || p.variant == 22 || p.variant == 24 || p.variant == 25 || p.variant == 29 || p.variant == 30;
    // Ring kernels issue the DMA as buffer loads: one SGPR resource per operand, a 32-bit per-lane offset fixed for
    // the tile and the k offset as the scalar offset -- no 64-bit address VALU per piece and half the address
    // traffic of the flat form (the piece's issue cost is what the K loop pays for).
    const long long a_bytes = (long long)(p.amap.rpg ? ((p.M + p.amap.rpg - 1) / p.amap.rpg) * (long long)p.amap.gstride + p.amap.off + p.amap.rpg
                                                      : p.M) * lda_b;
    const long long w_bytes = (long long)p.N * ldw_b;
    const bool use_buf = NSLOT == 3 && p.variant != 32 && a_bytes < 0xfffff000ll && w_bytes < 0xfffff000ll;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, use_buf ? (unsigned)a_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, use_buf ? (unsigned)w_bytes : 0u, 0x00020000);
    int a_vo[NI_A], w_vo[NI_W];
    if (use_buf) {
#pragma unroll
        for (int i = 0; i < NI_A; ++i) a_vo[i] = (int)(a_src[i] - (const char*)p.A);
#pragma unroll
        for (int i = 0; i < NI_W; ++i) w_vo[i] = (int)(w_src[i] - (const char*)p.W);
    }
    auto issue = [&](int kt, int buf) {
        if (dbg_noload) return;
        char* base = smem + buf * BUF + wave_dst;
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc does not know this builtin)
        if (use_buf) {
            const int ko = kt * ROWB;
#pragma unroll
            for (int i = 0; i < NI_A; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lptr_t)(base + i * NW * 1024), 16, a_vo[i], ko, 0, 0);
#pragma unroll
            for (int i = 0; i < NI_W; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lptr_t)(base + BM * ROWB + i * NW * 1024), 16, w_vo[i], ko, 0, 0);
            return;
        }
#endif
        const long long ko = (long long)kt * ROWB;
#pragma unroll
        for (int i = 0; i < NI_A; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + ko), (lptr_t)(base + i * NW * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NI_W; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + ko), (lptr_t)(base + BM * ROWB + i * NW * 1024), 16, 0, 0);
    };

    issue(0, 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // residual prefetch (same element order as the epilogue)
    // (only for wave tiles up to 64x64: a 128x64 wave tile already holds 128 accumulator registers and loads its
    // residual in the epilogue instead)
    constexpr bool RES_PREFETCH = (EPI & EPI_RES) != 0 && TM * TN <= 4;
    float rres[RES_PREFETCH ? TM * 16 * TN : 1];
    if constexpr (RES_PREFETCH) {
        if (p.cmap.rpg == 0 && row0 + BM <= p.M) {  // scalar row offsets, one per-lane offset (gemm_epilogue.h)
            const long long rb = (long long)p.M * p.ldr * 4;
            const __amdgpu_buffer_rsrc_t rrs =
                __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, rb > 0xfffff000ll ? 0xfffff000u : (unsigned)rb, 0x00020000);
            const int wu = __builtin_amdgcn_readfirstlane(wid);
            const int rbase = row0 + (wu / WN) * WTM, cbase = col0 + (wu % WN) * WTN;
            const int vo_r = (4 * (lane >> 5) * p.ldr + (lane & 31)) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int so_r = ((rbase + i * 32 + (reg & 3) + 8 * (reg >> 2)) * p.ldr + cbase) * 4;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        rres[(i * 16 + reg) * TN + j] =
                            __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, vo_r + j * 128, so_r, 0));
                }
        } else
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int r = row0 + wr * WTM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (r >= p.M) r = p.M - 1;
                const long long pr = map_row2(p.cmap, r);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    rres[(i * 16 + reg) * TN + j] = p.res[pr * p.ldr + col0 + wc * WTN + j * 32 + (lane & 31)];
            }
    }

    // fragment read offsets: row (lane&31), logical chunk 2s+h at position (2s+h) ^ ((row>>1)&7)
    const int l31 = lane & 31, lh = lane >> 5;
    const int sw = ROWB == 128 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = l31 * ROWB + (((2 * s + lh) ^ sw) * 16);
    const int fragA = wr * WTM * ROWB;
    const int fragW = BM * ROWB + wc * WTN * ROWB;

    if constexpr (NSLOT == 3) {
        constexpr int PER = NI_A + NI_W;  // DMA wave-instructions per stage
        // VMEM order so far: stage 0, (residual prefetch), now stage 1: "all but the newest PER" = stage kt landed
        if (nkt > 1) issue(1, 1);
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt)
                glds_wait_barrier<PER>();
            else
                glds_wait_barrier<0>();
            // every wave has finished reading slot (kt-1)%3 = (kt+2)%3 before it passed this barrier
            if (kt + 2 < nkt) issue(kt + 2, (kt + 2) % 3);
            const char* cur = smem + (kt % 3) * BUF;
            u32x4 fa[KS][TM], fw[KS][TN];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[s][i] = *(const u32x4*)(cur + fragA + i * 32 * ROWB + foff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fw[s][j] = *(const u32x4*)(cur + fragW + j * 32 * ROWB + foff[s]);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, fa[s][i]), __builtin_bit_cast(bf16x8, fw[s][j]), acc[i][j], 0, 0, 0);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }

    __syncthreads();  // (the compiler drains vmcnt before a barrier while LDS-DMA is outstanding)

    for (int kt = NSLOT == 3 ? nkt : 0; kt < nkt; ++kt) {
        const char* cur = smem + (kt & 1) * BUF;
        if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
        if (p.variant == 24 || p.variant == 25) {  // timing experiment: DMA stream + barriers only, no LDS reads / MFMA
            __syncthreads();
            continue;
        }
        if constexpr (TM == 4 && TN == 2 && ROWB == 128) {
            // Hand-scheduled stage (hipcc serialises this loop with lgkmcnt(0) after every few reads): two
            // fragment register sets; the 6 ds_read_b128 of k-step s+1 are in flight under the 8 MFMAs of
            // k-step s, waits are counted (lgkmcnt(6) = "all but the newest set").
            const unsigned lb = (unsigned)(size_t)(lptr_t)smem + (unsigned)((kt & 1) * BUF);
            const unsigned aA0 = lb + fragA + foff[0], aA1 = lb + fragA + foff[1], aA2 = lb + fragA + foff[2],
                           aA3 = lb + fragA + foff[3];
            const unsigned aW0 = lb + fragW + foff[0], aW1 = lb + fragW + foff[1], aW2 = lb + fragW + foff[2],
                           aW3 = lb + fragW + foff[3];
            u32x4 t0, t1, t2, t3, t4, t5, u0, u1, u2, u3, u4, u5;
            asm volatile(
                "ds_read_b128 %8, %20\n\t"
                "ds_read_b128 %9, %20 offset:4096\n\t"
                "ds_read_b128 %10, %20 offset:8192\n\t"
                "ds_read_b128 %11, %20 offset:12288\n\t"
                "ds_read_b128 %12, %24\n\t"
                "ds_read_b128 %13, %24 offset:4096\n\t"
                "ds_read_b128 %14, %21\n\t"
                "ds_read_b128 %15, %21 offset:4096\n\t"
                "ds_read_b128 %16, %21 offset:8192\n\t"
                "ds_read_b128 %17, %21 offset:12288\n\t"
                "ds_read_b128 %18, %25\n\t"
                "ds_read_b128 %19, %25 offset:4096\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %8, %12, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %8, %13, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %9, %12, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %9, %13, %3\n\t"
                "v_mfma_f32_32x32x16_bf16 %4, %10, %12, %4\n\t"
                "v_mfma_f32_32x32x16_bf16 %5, %10, %13, %5\n\t"
                "v_mfma_f32_32x32x16_bf16 %6, %11, %12, %6\n\t"
                "v_mfma_f32_32x32x16_bf16 %7, %11, %13, %7\n\t"
                "ds_read_b128 %8, %22\n\t"
                "ds_read_b128 %9, %22 offset:4096\n\t"
                "ds_read_b128 %10, %22 offset:8192\n\t"
                "ds_read_b128 %11, %22 offset:12288\n\t"
                "ds_read_b128 %12, %26\n\t"
                "ds_read_b128 %13, %26 offset:4096\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %14, %18, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %14, %19, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %15, %18, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %15, %19, %3\n\t"
                "v_mfma_f32_32x32x16_bf16 %4, %16, %18, %4\n\t"
                "v_mfma_f32_32x32x16_bf16 %5, %16, %19, %5\n\t"
                "v_mfma_f32_32x32x16_bf16 %6, %17, %18, %6\n\t"
                "v_mfma_f32_32x32x16_bf16 %7, %17, %19, %7\n\t"
                "ds_read_b128 %14, %23\n\t"
                "ds_read_b128 %15, %23 offset:4096\n\t"
                "ds_read_b128 %16, %23 offset:8192\n\t"
                "ds_read_b128 %17, %23 offset:12288\n\t"
                "ds_read_b128 %18, %27\n\t"
                "ds_read_b128 %19, %27 offset:4096\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %8, %12, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %8, %13, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %9, %12, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %9, %13, %3\n\t"
                "v_mfma_f32_32x32x16_bf16 %4, %10, %12, %4\n\t"
                "v_mfma_f32_32x32x16_bf16 %5, %10, %13, %5\n\t"
                "v_mfma_f32_32x32x16_bf16 %6, %11, %12, %6\n\t"
                "v_mfma_f32_32x32x16_bf16 %7, %11, %13, %7\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %14, %18, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %14, %19, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %15, %18, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %15, %19, %3\n\t"
                "v_mfma_f32_32x32x16_bf16 %4, %16, %18, %4\n\t"
                "v_mfma_f32_32x32x16_bf16 %5, %16, %19, %5\n\t"
                "v_mfma_f32_32x32x16_bf16 %6, %17, %18, %6\n\t"
                "v_mfma_f32_32x32x16_bf16 %7, %17, %19, %7\n\t"
                : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),
                  "+v"(acc[3][0]), "+v"(acc[3][1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5),
                  "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5)
                : "v"(aA0), "v"(aA1), "v"(aA2), "v"(aA3), "v"(aW0), "v"(aW1), "v"(aW2), "v"(aW3)
                : "memory");
            if (kt + 1 == nkt) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU readers
            __syncthreads();
            continue;
        }
        if constexpr (TM == 2 && TN == 2 && ROWB == 128) {
            // Same idea for the 64x64 wave tile: hipcc's schedule of this loop puts an lgkmcnt(0) in front of almost
            // every MFMA pair (8 exposed LDS round trips per stage); here the 4 reads of k-step s+1 fly under the 4
            // MFMAs of k-step s and the waits are counted.
            const unsigned lb = (unsigned)(size_t)(lptr_t)smem + (unsigned)((kt & 1) * BUF);
            const unsigned aA0 = lb + fragA + foff[0], aA1 = lb + fragA + foff[1], aA2 = lb + fragA + foff[2],
                           aA3 = lb + fragA + foff[3];
            const unsigned aW0 = lb + fragW + foff[0], aW1 = lb + fragW + foff[1], aW2 = lb + fragW + foff[2],
                           aW3 = lb + fragW + foff[3];
            u32x4 t0, t1, t2, t3, u0, u1, u2, u3;
            asm volatile(
                "ds_read_b128 %4, %12\n\t"
                "ds_read_b128 %6, %16\n\t"
                "ds_read_b128 %5, %12 offset:4096\n\t"
                "ds_read_b128 %7, %16 offset:4096\n\t"
                "ds_read_b128 %8, %13\n\t"
                "ds_read_b128 %10, %17\n\t"
                "ds_read_b128 %9, %13 offset:4096\n\t"
                "ds_read_b128 %11, %17 offset:4096\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %4, %6, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %7, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %5, %6, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %5, %7, %3\n\t"
                "ds_read_b128 %4, %14\n\t"
                "ds_read_b128 %6, %18\n\t"
                "ds_read_b128 %5, %14 offset:4096\n\t"
                "ds_read_b128 %7, %18 offset:4096\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %8, %10, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %8, %11, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %9, %10, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %9, %11, %3\n\t"
                "ds_read_b128 %8, %15\n\t"
                "ds_read_b128 %10, %19\n\t"
                "ds_read_b128 %9, %15 offset:4096\n\t"
                "ds_read_b128 %11, %19 offset:4096\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %4, %6, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %7, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %5, %6, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %5, %7, %3\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %8, %10, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %8, %11, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %9, %10, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %9, %11, %3\n\t"
                : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "=&v"(t0), "=&v"(t1), "=&v"(t2),
                  "=&v"(t3), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
                : "v"(aA0), "v"(aA1), "v"(aA2), "v"(aA3), "v"(aW0), "v"(aW1), "v"(aW2), "v"(aW3)
                : "memory");
            if (kt + 1 == nkt) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU readers
            __syncthreads();
            continue;
        }
        // software-pipelined fragments: the ds_reads of k-step s+1 are issued before the MFMAs of k-step s,
        // one read per MFMA slot (sched_group_barrier), so LDS latency hides under the matrix pipe
        u32x4 fa[2][TM], fw[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *(const u32x4*)(cur + fragA + i * 32 * ROWB + foff[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fw[0][j] = *(const u32x4*)(cur + fragW + j * 32 * ROWB + foff[0]);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s < KS - 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[(s + 1) & 1][i] = *(const u32x4*)(cur + fragA + i * 32 * ROWB + foff[s + 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fw[(s + 1) & 1][j] = *(const u32x4*)(cur + fragW + j * 32 * ROWB + foff[s + 1]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, fa[s & 1][i]), __builtin_bit_cast(bf16x8, fw[s & 1][j]), acc[i][j], 0, 0, 0);
            if (s < KS - 1) {
#pragma unroll
                for (int x = 0; x < TM + TN; ++x) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
                if constexpr (TM * TN > TM + TN) __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
            }
        }
        __syncthreads();
    }

    float bj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bj[j] = p.bias ? p.bias[col0 + wc * WTN + j * 32 + l31] : 0.f;
    if (dbg_nostore) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 12345.678f) p.Cf[0] = t;
        return;
    }
    if (p.cmap.rpg == 0 && row0 + BM <= p.M && !(EPI & EPI_ROWTAB)) {
        // full tile, identity row map: scalar row offset (buffer soffset) + one per-lane offset (voffset);
        // see gemm_epilogue.h.  The residual comes from the registers prefetched before the K loop.
        constexpr int ES = (EPI & EPI_F32OUT) ? 4 : 2;
        void* cptr = (EPI & EPI_F32OUT) ? (void*)p.Cf : (void*)p.Cb;
        const long long cb = (long long)p.M * p.ldc * ES;
        const __amdgpu_buffer_rsrc_t crs =
            __builtin_amdgcn_make_buffer_rsrc(cptr, 0, cb > 0xfffff000ll ? 0xfffff000u : (unsigned)cb, 0x00020000);
        const int wu = __builtin_amdgcn_readfirstlane(wid);
        const int rbase = row0 + (wu / WN) * WTM, cbase = col0 + (wu % WN) * WTN;
        const int vo_c = (4 * lh * p.ldc + l31) * ES;
        const long long rb2 = (long long)p.M * p.ldr * 4;
        const __amdgpu_buffer_rsrc_t rrs2 = __builtin_amdgcn_make_buffer_rsrc(
            (EPI & EPI_RES) ? (void*)p.res : cptr, 0, rb2 > 0xfffff000ll ? 0xfffff000u : (unsigned)rb2, 0x00020000);
        const int vo_r2 = (4 * lh * p.ldr + l31) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int so_c = ((rbase + i * 32 + (reg & 3) + 8 * (reg >> 2)) * p.ldc + cbase) * ES;
                const int so_r2 = ((rbase + i * 32 + (reg & 3) + 8 * (reg >> 2)) * p.ldr + cbase) * 4;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v = acc[i][j][reg] + bj[j];
                    if constexpr (EPI & EPI_GELU) v = gelu_fast2(v);
                    if constexpr (RES_PREFETCH)
                        v += rres[(i * 16 + reg) * TN + j];
                    else if constexpr (EPI & EPI_RES)
                        v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs2, vo_r2 + j * 128, so_r2, 0));
                    if constexpr (EPI & EPI_F32OUT)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, vo_c + j * 128, so_c, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)v), crs, vo_c + j * 64,
                                                              so_c, 0);
                }
            }
        }
        return;
    }
    // General path (row-mapped outputs, row tables, partial tiles).  The row map and the row-table index need an
    // integer division per ROW; done per lane and element they cost more than the tile's MFMAs, so one thread per tile
    // row does them once and parks {physical row | -1, row-table row} in LDS (free after the K loop).
    __syncthreads();
    int* rowinfo = (int*)smem;
    for (int rr = tid; rr < BM; rr += NW * 64) {
        const int gr = row0 + rr;
        rowinfo[2 * rr] = gr < p.M ? map_row2(p.cmap, gr) : -1;
        rowinfo[2 * rr + 1] = (EPI & EPI_ROWTAB) ? gr % p.rt_mod : 0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rl = wr * WTM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const int pri = rowinfo[2 * rl], tri = rowinfo[2 * rl + 1];
            if (pri >= 0) {
                const long long pr = pri;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int c = col0 + wc * WTN + j * 32 + l31;
                    float v = acc[i][j][reg] + bj[j];
                    if constexpr (EPI & EPI_ROWTAB) v += p.rowtab[tri * p.rt_ld + c];
                    if constexpr (EPI & EPI_GELU) v = gelu_fast2(v);
                    if constexpr (RES_PREFETCH)
                        v += rres[(i * 16 + reg) * TN + j];
                    else if constexpr (EPI & EPI_RES)
                        v += p.res[pr * p.ldr + c];
                    if constexpr (EPI & EPI_F32OUT)
                        p.Cf[pr * p.ldc + c] = v;
                    else
                        p.Cb[pr * p.ldc + c] = (bf16_t)v;
                }
            }
        }
    }
}

// Row peeling (PEEL): the chip holds 512 resident 128x128 workgroups, and a grid of e.g. 1568 tiles runs as 3.06
// "rounds", i.e. costs 4.  With p.peel = P (a multiple of BM chosen by the launcher so that the tiles of rows
// [0, P) fill whole rounds) the rows [P, M) are covered by 64x64 tiles whose workgroups come FIRST in the grid: they
// are dispatched at t = 0 beside the first big tiles, take a quarter of a big tile's time, and the grid ends after
// ~3.1 rounds.  Every output element still accumulates its k-steps in the same order, so results do not depend on
// the tile shape.
template <int BM, int BN, int EPI, int WM, int WN, int ROWB, bool PEEL>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_glds_kernel(GemmP p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * (BM + BN) * ROWB];
    if constexpr (PEEL) {
        const int ntm_tail = (p.M - p.peel + 63) / 64;
        const int n_tail = ntm_tail * (p.N / 64);
        if ((int)blockIdx.x < n_tail) {
            gemm_glds_tile<64, 64, EPI, 2, 2, ROWB>(p, smem, blockIdx.x, ntm_tail, p.peel);
            return;
        }
        gemm_glds_tile<BM, BN, EPI, WM, WN, ROWB>(p, smem, blockIdx.x - n_tail, p.peel / BM, 0);
    } else {
        gemm_glds_tile<BM, BN, EPI, WM, WN, ROWB>(p, smem, blockIdx.x, (p.M + BM - 1) / BM, 0);
    }
}

// 128x128 tile on the three-slot ring (NSLOT = 3): 48 KiB of LDS and <= 168 VGPRs => three workgroups per CU
template <int EPI>
__global__ __launch_bounds__(256, 3) void gemm_glds_ring3_kernel(GemmP p) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * (128 + 128) * 64];
    if (p.peel > 0) {  // rows >= peel on 64x64 tiles whose workgroups come first in the grid (see gemm_glds_kernel)
        const int ntm_tail = (p.M - p.peel + 63) / 64;
        const int n_tail = ntm_tail * (p.N / 64);
        if ((int)blockIdx.x < n_tail) {
            gemm_glds_tile<64, 64, EPI, 2, 2, 64, 3>(p, smem, blockIdx.x, ntm_tail, p.peel);
            return;
        }
        gemm_glds_tile<128, 128, EPI, 2, 2, 64, 3>(p, smem, blockIdx.x - n_tail, p.peel / 128, 0);
        return;
    }
    // clock probe (variants 26/28-30, tools/gemm_bench.py --clock): shader-clock and 100-MHz wall counters around
    // one mid-grid workgroup give the SIMD clock this kernel actually ran at
    const bool probe = p.variant >= 26 && p.variant <= 30 && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
    long long c0 = 0, w0 = 0;
    if (probe) {
        c0 = clock64();
        w0 = wall_clock64();
    }
    gemm_glds_tile<128, 128, EPI, 2, 2, 64, 3>(p, smem, blockIdx.x, (p.M + 127) / 128, 0);
    if (probe) {
        g_clock_probe[0] = clock64() - c0;
        g_clock_probe[1] = wall_clock64() - w0;
    }
}
// experimental (variant 31): 256x128 tile, 4 waves with 128x64 wave tiles, same ring => 72 KiB, two workgroups per
// CU, 0.75x the L2->LDS bytes per flop of the 128x128 tile.  Epilogues without a residual only.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_glds_ring3w_kernel(GemmP p) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * (256 + 128) * 64];
    gemm_glds_tile<256, 128, EPI, 2, 2, 64, 3>(p, smem, blockIdx.x, (p.M + 255) / 256, 0);
}
static bool launch_ring3w(const GemmP& p, hipStream_t st) {
    const bool f32out = p.Cf != nullptr;
    const int epi = (p.gelu ? EPI_GELU : 0) | (p.res ? EPI_RES : 0) | (p.rowtab ? EPI_ROWTAB : 0) | (f32out ? EPI_F32OUT : 0);
    const dim3 grid(((p.M + 255) / 256) * (p.N / 128)), block(256);
    switch (epi) {
        case 0: hipLaunchKernelGGL(gemm_glds_ring3w_kernel<0>, grid, block, 0, st, p); return true;
        case EPI_GELU: hipLaunchKernelGGL(gemm_glds_ring3w_kernel<EPI_GELU>, grid, block, 0, st, p); return true;
        default: return false;
    }
}

static bool launch_ring3(const GemmP& p, hipStream_t st) {
    const bool f32out = p.Cf != nullptr;
    const int epi = (p.gelu ? EPI_GELU : 0) | (p.res ? EPI_RES : 0) | (p.rowtab ? EPI_ROWTAB : 0) | (f32out ? EPI_F32OUT : 0);
    const int tiles = p.peel > 0 ? (p.peel / 128) * (p.N / 128) + ((p.M - p.peel + 63) / 64) * (p.N / 64)
                                 : ((p.M + 127) / 128) * (p.N / 128);
    const dim3 grid(tiles), block(256);
    switch (epi) {
        case 0: hipLaunchKernelGGL(gemm_glds_ring3_kernel<0>, grid, block, 0, st, p); return true;
        case EPI_F32OUT: hipLaunchKernelGGL(gemm_glds_ring3_kernel<EPI_F32OUT>, grid, block, 0, st, p); return true;
        case EPI_GELU: hipLaunchKernelGGL(gemm_glds_ring3_kernel<EPI_GELU>, grid, block, 0, st, p); return true;
        case EPI_GELU | EPI_F32OUT: hipLaunchKernelGGL(gemm_glds_ring3_kernel<EPI_GELU | EPI_F32OUT>, grid, block, 0, st, p); return true;
        case EPI_RES | EPI_F32OUT: hipLaunchKernelGGL(gemm_glds_ring3_kernel<EPI_RES | EPI_F32OUT>, grid, block, 0, st, p); return true;
        case EPI_ROWTAB | EPI_F32OUT: hipLaunchKernelGGL(gemm_glds_ring3_kernel<EPI_ROWTAB | EPI_F32OUT>, grid, block, 0, st, p); return true;
        default: return false;
    }
}
// p.peel for `slots` resident 128x128 workgroups: the largest row count whose tiles fill whole rounds, when the
// remaining partial round would be less than half full and the peeled rows are few (else 0)
static int peel_rows(const GemmP& p, int slots) {
    const int ntm = (p.M + 127) / 128, ntn = p.N / 128;
    int g = slots, b = ntn;
    while (b) {
        const int t = g % b;
        g = b;
        b = t;
    }
    const int step = slots / g;  // row tiles per whole number of rounds
    const int ntm_main = (ntm / step) * step;
    const long long rem = ((long long)ntm * ntn) % slots;
    if (ntm_main > 0 && rem != 0 && rem * 2 < slots && p.M - ntm_main * 128 <= 4096) return ntm_main * 128;
    return 0;
}

template <int BM, int BN, int EPI, int WM, int WN, int ROWB = 128>
static void launch_cfg(const GemmP& p, hipStream_t st) {
    if (p.peel > 0) {
        if constexpr (BM == 128 && BN == 128 && WM == 2 && WN == 2) {
            const int grid = (p.peel / BM) * (p.N / BN) + ((p.M - p.peel + 63) / 64) * (p.N / 64);
            hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, EPI, WM, WN, ROWB, true>), dim3(grid), dim3(256), 0, st, p);
            return;
        }
    }
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
    hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, EPI, WM, WN, ROWB, false>), dim3(grid), dim3(WM * WN * 64), 0, st, p);
}

template <int BM, int BN, int WM, int WN, int ROWB = 128>
static bool launch_tile(const GemmP& p, hipStream_t st) {
    const bool f32out = p.Cf != nullptr;
    const int epi = (p.gelu ? EPI_GELU : 0) | (p.res ? EPI_RES : 0) | (p.rowtab ? EPI_ROWTAB : 0) | (f32out ? EPI_F32OUT : 0);
    switch (epi) {
        case 0: launch_cfg<BM, BN, 0, WM, WN, ROWB>(p, st); return true;
        case EPI_F32OUT: launch_cfg<BM, BN, EPI_F32OUT, WM, WN, ROWB>(p, st); return true;
        case EPI_GELU: launch_cfg<BM, BN, EPI_GELU, WM, WN, ROWB>(p, st); return true;
        case EPI_GELU | EPI_F32OUT: launch_cfg<BM, BN, EPI_GELU | EPI_F32OUT, WM, WN, ROWB>(p, st); return true;
        case EPI_RES | EPI_F32OUT: launch_cfg<BM, BN, EPI_RES | EPI_F32OUT, WM, WN, ROWB>(p, st); return true;
        case EPI_ROWTAB | EPI_F32OUT: launch_cfg<BM, BN, EPI_ROWTAB | EPI_F32OUT, WM, WN, ROWB>(p, st); return true;
        default: return false;
    }
}

// returns false when the shape / epilogue is not covered (caller falls back to gemm.hip's kernel)
bool launch_gemm_glds(const GemmP& p, hipStream_t st) {
    if (p.K % 64 != 0 || p.N % 128 != 0) return false;
    const long long big_tiles = (long long)((p.M + 127) / 128) * (p.N / 128);
    if (big_tiles < 512) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 8) || (p.ldw % 8)) return false;
    // (the 256-row double-buffer tilings, former variants 4-6 / 25, were retired: after this round's refactors they
    // spill under their 256-register budget, and they never beat the 128x128 tile on these shapes)
    if (p.variant == 3) return launch_tile<128, 128, 2, 2>(p, st);
    if (p.variant == 2) return launch_tile<128, 128, 2, 2, 64>(p, st);
    if ((p.variant >= 20 && p.variant <= 22) || p.variant == 24) return launch_tile<128, 128, 2, 2>(p, st);
    GemmP q = p;
    q.peel = 0;
    if (p.variant == 26 || (p.variant >= 28 && p.variant <= 30) || p.variant == 32) return launch_ring3(q, st);  // ring, no peeling (32: flat-address DMA)
    if (p.variant == 31) return launch_ring3w(q, st) || launch_ring3(q, st);
    if (p.variant == 23 || p.variant == 27) {         // double buffer, two workgroups per CU (23: no peeling)
        if (p.variant == 27) q.peel = peel_rows(p, 512);
        return launch_tile<128, 128, 2, 2>(q, st);
    }
    // default selection (measured with tools/gemm_bench.py on the plan step's shapes): 128x128 tiles on the
    // three-slot ring, three workgroups per CU -- two 16-KiB stages in flight per workgroup against the DMA latency,
    // and one workgroup's epilogue (VALU + stores) runs beside the others' MFMAs.  -9 % over the double buffer.
    q.peel = peel_rows(p, 768);
    if (launch_ring3(q, st)) return true;
    q.peel = peel_rows(p, 512);
    return launch_tile<128, 128, 2, 2>(q, st);
}

}  // namespace m3pc
