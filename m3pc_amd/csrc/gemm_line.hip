// bf16 MFMA GEMM whose LDS-DMA pieces are WHOLE 128-byte cache lines (GemmP::variant 43: 128x128 tiles, two
// workgroups per CU; 44: 256x256 tiles, one workgroup per CU).
//
// Why: the texture path prices a DMA wave-instruction by the cache lines it touches, not by its bytes.  A piece of 16
// rows x 64 B (the 32-deep stages of gemm_glds.hip / gemm_big.hip) touches 16 lines and uses half of each; 8 rows x
// 128 B moves the same KiB through 8.  Measured on this kernel's own stream (gemm_big.hip DBG 5/6): 2 MiB per CU in
// 22.4 us against 39.6 us.  Whole lines mean 64-deep K "pairs", i.e. 128-byte LDS rows, and twice the LDS per unit of
// prefetch; what makes that fit is a ring of FIVE half-stage units (A or W rows of one pair, BT x 128 B each):
//      unit 2u = A(u), unit 2u+1 = W(u), unit n lives in slot n % 5
//   resident while pair u is multiplied:  A(u) W(u) | A(u+1) W(u+1) A(u+2)   -- A (streamed from HBM) runs two pairs
//   ahead, W (L2-resident) one.  5 x 16 KiB = 80 KiB for 128x128 tiles (two workgroups per CU), 5 x 32 KiB = all
//   160 KiB for 256x256.
// Loop per pair u (four k-steps of 16; T*T MFMAs, 2T ds_read_b128 and T DMA pieces per wave each, interleaved by
// sched_group_barrier; two fragment register sets, the reads of k-step t+1 run under the MFMAs of k-step t):
//   k-step 0: second half of W(u+1)      k-step 1, 2: A(u+2)
//   k-step 3: s_waitcnt vmcnt(PU) lgkmcnt(0) + s_barrier -- everything but the A(u+2) pieces just issued has landed,
//             i.e. A(u+1) and W(u+1) everywhere, and every wave is past its last read of pair u -- then the first half
//             of W(u+2) into the slot A(u) just vacated, and the first fragments of pair u+1.
// LDS rows are 128 B; chunk position c of row r holds logical chunk c ^ ((r >> 1) & 7), applied on the global source
// address (the DMA writes LDS linearly).  Same MFMA instruction, k order and epilogue arithmetic as the other bf16
// kernels: results are bit-identical (tests/test_gemm_kernels_gpu.py).
#include <type_traits>

#include "gemm_epilogue.h"
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) * lptr_t;

template <int PU>
__device__ __forceinline__ void line_wait_barrier() {
    static_assert(PU == 4 || PU == 8 || PU == 6 || PU == 12 || PU == 16 || PU == 24, "add the immediate");
    if constexpr (PU == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (PU == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (PU == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (PU == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (PU == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (PU == 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DBG (timing experiments, tools/gemm_bench.py 45/46): 1 = no DMA pieces, 2 = no MFMA
template <int BT, int EPI, int DBG = 0>
__global__ __launch_bounds__(256, BT == 128 ? 2 : 1) void gemm_line_kernel(GemmP p) {
    constexpr int U = BT * 128;   // bytes per ring unit: BT rows x 128 B
    constexpr int T = BT / 64;    // 32x32 tiles per wave per dimension (wave tile BT/2 x BT/2)
    constexpr int PU = BT / 32;   // DMA pieces (8 rows x 128 B) per wave per unit
    constexpr int G = 2 * T;      // fragment reads per k-step (T of A, T of W)
    constexpr int MF = T * T / G; // MFMAs per read
    __shared__ __attribute__((aligned(1024))) char smem[5 * U];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int ntn = p.N / BT, ntm = (p.M + BT - 1) / BT, nwg = ntm * ntn;
    const int lda_b = p.lda * 2, ldw_b = p.ldw * 2;
    const int np = p.K / 64;  // pairs
    const long long a_rows = p.amap.rpg ? ((p.M + p.amap.rpg - 1) / p.amap.rpg) * (long long)p.amap.gstride + p.amap.off + p.amap.rpg : p.M;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)(a_rows * lda_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((long long)p.N * ldw_b), 0x00020000);

    // Persistent workgroups: workgroup b multiplies tiles b, b + gridDim.x, ...; the ring runs on ACROSS tiles -- the
    // pieces issued "past the end" of a tile are the first pairs of the next one, so only a workgroup's first tile pays
    // a prologue.  Tile order: consecutive workgroups (same XCD) share A rows.
    auto tile_origin = [&](int t, int& r0, int& c0) {
        const int q = nwg / 8, r = nwg % 8, x = t % 8, i = t / 8;
        const int b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        r0 = (b / ntn) * BT;
        c0 = (b % ntn) * BT;
    };
    // DMA pieces: piece I = wid + 4 i (i < PU) of a unit = its rows 8 I .. 8 I + 7, 8 lanes per 128-byte row.  The
    // per-lane source offsets are relative to the tile, the tile itself is a scalar byte offset (buffer soffset) -- so
    // moving on to the next tile costs no registers.  That needs every tile to look alike: no A row map, and a ragged
    // last tile only when the caller vouches for readable memory behind A (GemmP::a_padded): the tile's origin rides in
    // the buffer instruction's scalar offset, which the hardware range check does NOT include, so the DMA of a ragged
    // tile reads up to 127 rows past M (never stored).  Otherwise (`uniform` false) the launcher starts one workgroup
    // per tile and the per-lane offsets are absolute and clamped to row M - 1.
    const bool uniform = p.amap.rpg == 0 && (p.M % BT == 0 || p.a_padded);
    int a_vo[PU], w_vo[PU];
    auto piece_offsets = [&](int r0, int c0) {
#pragma unroll
        for (int i = 0; i < PU; ++i) {
            const int r = 8 * (wid + 4 * i) + (lane >> 3);
            const int q = (lane & 7) ^ ((r >> 1) & 7);
            if (uniform) {
                a_vo[i] = r * lda_b + q * 16;
                w_vo[i] = r * ldw_b + q * 16;
            } else {
                int gr = r0 + r;
                if (gr >= p.M) gr = p.M - 1;
                a_vo[i] = ge_map_row(p.amap, gr) * lda_b + q * 16;
                w_vo[i] = (c0 + r) * ldw_b + q * 16;
            }
        }
    };
    const int wave_dst = __builtin_amdgcn_readfirstlane(wid) * 1024;
    (void)a_rs, (void)w_rs, (void)wave_dst;  // (only the device pass uses them: the DMA builtin is hidden from the host pass)
    // piece i of the A (w = false) or W (w = true) rows at scalar byte offset `so` (tile + pair) into ring slot `slot`
    auto piece = [&](int slot, int so, bool w, int i) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc does not know this builtin)
        if (DBG == 1) return;
        char* dst = smem + slot * U + wave_dst + i * 4096;
        if (!w)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lptr_t)dst, 16, a_vo[i], so, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lptr_t)dst, 16, w_vo[i], so, 0, 0);
#endif
    };

    const int sw = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) foff[t] = l31 * 128 + (((2 * t + lh) ^ sw) * 16);
    const int fragA = wr * (BT / 2) * 128, fragW = wc * (BT / 2) * 128;
    u32x4 fa[2][T], fw[2][T];  // two fragment register sets: k-step t multiplies set t & 1 while the other is being read
    // one fragment read of k-step t: g < T -> A row tile g (unit in slot sa), else W row tile g - T (slot sw_)
    auto frag = [&](int set, int sa, int sw_, int t, int g) {
        if (g < T)
            fa[set][g] = *(const u32x4*)(smem + sa * U + fragA + g * 4096 + foff[t]);
        else
            fw[set][g - T] = *(const u32x4*)(smem + sw_ * U + fragW + (g - T) * 4096 + foff[t]);
    };
    // one k-step's worth of scheduling groups: MF MFMAs + 1 read, a DMA piece after every second read
    auto pattern = [&](bool dma) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            if (dma && (g & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (DMA piece)
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    int tile = blockIdx.x;
    int row0, col0;
    tile_origin(tile, row0, col0);
    piece_offsets(row0, col0);
    int a_base = uniform ? row0 * lda_b : 0, w_base = uniform ? col0 * ldw_b : 0;  // this tile's scalar offsets

    // prologue (first tile only): A(0) W(0) A(1) and the first half of W(1) in flight, A(0), W(0) landed, the fragments
    // of k-step 0 read -- the state every later tile finds when its predecessor's loop ends.  (np >= 3: the launcher checks)
#pragma unroll
    for (int i = 0; i < PU; ++i) piece(0, a_base, false, i);
#pragma unroll
    for (int i = 0; i < PU; ++i) piece(1, w_base, true, i);
#pragma unroll
    for (int i = 0; i < PU; ++i) piece(2, a_base + 128, false, i);
#pragma unroll
    for (int i = 0; i < PU / 2; ++i) piece(3, w_base + 128, true, i);
    line_wait_barrier<PU + PU / 2>();
#pragma unroll
    for (int g = 0; g < G; ++g) frag(0, 0, 1, 0, g);

    int a0 = 0;  // slot of A(u)
    auto mma = [&](f32x16 (&acc)[T][T], int set) {
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j)
                if (DBG == 2) {
                    if (i == j) acc[i][j][0] += __builtin_bit_cast(float, fa[set][i][0] ^ fw[set][j][1]);
                } else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][i]),
                                                                    __builtin_bit_cast(bf16x8, fw[set][j]), acc[i][j], 0, 0, 0);
    };
    // One pair of the K loop on the accumulators `acc`; `mid()` runs right after the pair's barrier.
    auto pair = [&](int u, int npairs, f32x16 (&acc)[T][T], int a_base_n, int w_base_n, auto&& mid) {
        // slots of A(u) W(u) A(u+1) W(u+1) A(u+2).  Pairs u+1, u+2 past the end of K are pairs 0, 1 of the next tile
        // (selected arithmetically: the body stays ONE basic block and the vmcnt immediate a constant).
        const int s_a = a0, s_w = a0 + 1 >= 5 ? a0 - 4 : a0 + 1, s_a1 = a0 + 2 >= 5 ? a0 - 3 : a0 + 2,
                  s_w1 = a0 + 3 >= 5 ? a0 - 2 : a0 + 3, s_a2 = a0 + 4 >= 5 ? a0 - 1 : a0 + 4;
        const bool nx1 = u + 1 >= npairs, nx2 = u + 2 >= npairs;
        const int so_w1 = (nx1 ? w_base_n + (u + 1 - npairs) * 128 : w_base + (u + 1) * 128);
        const int so_a2 = (nx2 ? a_base_n + (u + 2 - npairs) * 128 : a_base + (u + 2) * 128);
        const int so_w2 = (nx2 ? w_base_n + (u + 2 - npairs) * 128 : w_base + (u + 2) * 128);
        // (reads and DMA pieces alternate in SOURCE order: the compiler must assume they alias, so it keeps that order)
        // k-step 0: second half of W(u+1)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            frag(1, s_a, s_w, 1, g);
            if (g & 1) piece(s_w1, so_w1, true, PU / 2 + (g >> 1));
        }
        mma(acc, 0);
        pattern(true);
        // k-step 1: first half of A(u+2) into the slot W(u-1) left at the last barrier
#pragma unroll
        for (int g = 0; g < G; ++g) {
            frag(0, s_a, s_w, 2, g);
            if (g & 1) piece(s_a2, so_a2, false, g >> 1);
        }
        mma(acc, 1);
        pattern(true);
        // k-step 2: second half of A(u+2)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            frag(1, s_a, s_w, 3, g);
            if (g & 1) piece(s_a2, so_a2, false, PU / 2 + (g >> 1));
        }
        mma(acc, 0);
        pattern(true);
        // k-step 3: every read of pair u is back and A(u+1), W(u+1) have landed everywhere -> first fragments of pair
        // u+1; first half of W(u+2) into the slot A(u) vacated
        line_wait_barrier<PU>();
        mid();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            frag(0, s_a1, s_w1, 0, g);
            if (g & 1) piece(s_a, so_w2, true, g >> 1);
        }
        mma(acc, 1);
        pattern(true);
        a0 = s_a1;
    };
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    while (true) {
        const int tile_n = tile + (int)gridDim.x;
        const bool has_next = tile_n < nwg;
        int row_n = row0, col_n = col0;
        if (has_next) tile_origin(tile_n, row_n, col_n);
        // the next tile's scalar offsets (this tile's again after the last one: harmless re-reads of its first pairs)
        const int a_base_n = uniform ? row_n * lda_b : 0, w_base_n = uniform ? col_n * ldw_b : 0;

        // Everything the epilogue needs from memory is requested HERE, ahead of this tile's pieces: a load issued in the
        // epilogue would queue behind the next tile's DMA.
        // Residual epilogue (out = acc + bias + res, fp32) on full tiles with the identity row map: 32-row strips of the
        // wave tile, 16 T loads per strip.  128x128: both strips are fetched now (64 registers); 256x256: strip i+1 is
        // fetched while strip i is added and stored (the accumulators already take half the register file).
        constexpr bool RESEPI = EPI == (GE_RES | GE_F32OUT);
        constexpr bool PRE = RESEPI && BT == 128;
        const bool fastepi = RESEPI && p.cmap.rpg == 0 && row0 + BT <= p.M;
        const int rbase = row0 + (wu >> 1) * (BT / 2), cbase = col0 + (wu & 1) * (BT / 2);
        const __amdgpu_buffer_rsrc_t crs =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.Cf, 0, ge_clamp_bytes((long long)p.M * p.ldc * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rrs =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, ge_clamp_bytes((long long)p.M * p.ldr * 4), 0x00020000);
        const int vo_c = (4 * lh * p.ldc + l31) * 4, vo_r = (4 * lh * p.ldr + l31) * 4;
        constexpr int RS = 16 * T;  // residual values per lane per strip
        float R0[RS], R1[RS];
        auto load_strip = [&](float (&R)[RS], int i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int so_r = ((rbase + i * 32 + (reg & 3) + 8 * (reg >> 2)) * p.ldr + cbase) * 4;
#pragma unroll
                for (int j = 0; j < T; ++j)
                    R[reg * T + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, vo_r + j * 128, so_r, 0));
            }
        };
        float bj[T];
#pragma unroll
        for (int j = 0; j < T; ++j) bj[j] = p.bias ? p.bias[cbase + j * 32 + l31] : 0.f;
        if constexpr (PRE) {
            if (fastepi) {
                load_strip(R0, 0);
                load_strip(R1, 1);
            }
        }

        f32x16 acc[T][T];
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int u = 0; u < np; ++u) pair(u, np, acc, a_base_n, w_base_n, [] {});
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (MFMA results are read by the epilogue's VALU right away)

        bool done = false;
        if constexpr (RESEPI) {
            if (fastepi) {
                auto put_strip = [&](const float (&R)[RS], int i) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int so_c = ((rbase + i * 32 + (reg & 3) + 8 * (reg >> 2)) * p.ldc + cbase) * 4;
#pragma unroll
                        for (int j = 0; j < T; ++j) {
                            const float v = acc[i][j][reg] + bj[j] + R[reg * T + j];
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, vo_c + j * 128, so_c, 0);
                        }
                    }
                };
                if constexpr (PRE) {
                    put_strip(R0, 0);
                    put_strip(R1, 1);
                } else {
                    load_strip(R0, 0);
#pragma unroll
                    for (int i = 0; i < T; i += 2) {
                        load_strip(R1, i + 1);
                        put_strip(R0, i);
                        if (i + 2 < T) load_strip(R0, i + 2);
                        put_strip(R1, i + 1);
                    }
                }
                done = true;
            }
        }
        if (!done) gemm_epilogue<EPI, T, T>(p, acc, rbase, cbase, row0, BT, lane, bj);

        if (!has_next) break;
        tile = tile_n;
        row0 = row_n;
        col0 = col_n;
        a_base = a_base_n;
        w_base = w_base_n;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // drain the pieces issued past the last tile
}

template <int BT>
static bool launch_line(const GemmP& p, hipStream_t st) {
    const bool f32out = p.Cf != nullptr;
    const int epi = (p.gelu ? GE_GELU : 0) | (p.res ? GE_RES : 0) | (f32out ? GE_F32OUT : 0);
    const int tiles = ((p.M + BT - 1) / BT) * (p.N / BT), slots = (BT == 128 ? 2 : 1) * 256;  // resident workgroups on 256 CUs
    const bool uniform = p.amap.rpg == 0 && (p.M % BT == 0 || p.a_padded);  // (see the kernel: persistent workgroups need look-alike tiles)
    const dim3 grid(uniform && tiles > slots ? slots : tiles), block(256);
    switch (epi) {
        case 0:
            if (p.variant == 45) hipLaunchKernelGGL((gemm_line_kernel<BT, 0, 1>), grid, block, 0, st, p);
            else if (p.variant == 46) hipLaunchKernelGGL((gemm_line_kernel<BT, 0, 2>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((gemm_line_kernel<BT, 0, 0>), grid, block, 0, st, p);
            return true;
        case GE_F32OUT: hipLaunchKernelGGL((gemm_line_kernel<BT, GE_F32OUT>), grid, block, 0, st, p); return true;
        case GE_GELU: hipLaunchKernelGGL((gemm_line_kernel<BT, GE_GELU>), grid, block, 0, st, p); return true;
        case GE_GELU | GE_F32OUT: hipLaunchKernelGGL((gemm_line_kernel<BT, GE_GELU | GE_F32OUT>), grid, block, 0, st, p); return true;
        case GE_RES | GE_F32OUT: hipLaunchKernelGGL((gemm_line_kernel<BT, GE_RES | GE_F32OUT>), grid, block, 0, st, p); return true;
        default: return false;
    }
}

// returns false when the shape / epilogue is not covered (output row maps and row tables stay on gemm_glds.hip)
bool launch_gemm_line(const GemmP& p, int bt, hipStream_t st) {
    if (p.K % 64 != 0 || p.K < 192 || p.N % bt != 0 || p.rowtab || p.cmap.rpg != 0) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 8) || (p.ldw % 8)) return false;
    const long long a_rows = p.amap.rpg ? ((p.M + p.amap.rpg - 1) / p.amap.rpg) * (long long)p.amap.gstride + p.amap.off + p.amap.rpg : p.M;
    if (a_rows * p.lda * 2 >= 0x7ffff000ll || (long long)p.N * p.ldw * 2 >= 0x7ffff000ll) return false;
    return bt == 256 ? launch_line<256>(p, st) : launch_line<128>(p, st);
}

}  // namespace m3pc
