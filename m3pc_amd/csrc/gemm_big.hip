// bf16 MFMA GEMM for the long-K (K >= 1024) many-row problems: 256x256 block tile, FOUR waves (2 x 2) of 128x128 --
// one wave per SIMD, 256 accumulator registers each -- K streamed through a 4-slot LDS ring of 32-deep stages filled
// by buffer-form LDS-DMA.  (GemmP::variant 37 while it is being measured; see DESIGN.md for why this shape: a 128x128
// tile needs 2x the L2->LDS bytes per flop and this step's K = 2048 GEMMs are bound by exactly that stream.)
//
// With one wave per SIMD nothing hides an exposed wait, so the loop is software-pipelined in the source and pinned
// with sched_group_barrier: the 8 ds_read_b128 of k-step t+1 and the DMA pieces of stage kt+3 are interleaved with the
// 16 MFMAs of k-step t (two fragment register sets).  Per stage kt:
//   k-step 0:  reads (kt, k-step 1) -> set 1        under  MFMAs on set 0
//   k-step 1:  s_waitcnt vmcnt(8) + s_barrier  (stage kt+1 landed everywhere; every wave is past its reads of stage
//              kt-1) | DMA stage kt+3 -> slot (kt-1)%4 | reads (kt+1, k-step 0) -> set 0   under  MFMAs on set 1
// LDS rows are 64 B; chunk position c of row r holds logical chunk c ^ ((r >> 2) & 3), applied on the global source
// address (the DMA writes LDS linearly), which makes the ds_read_b128 groups conflict-free (as in gemm_glds.hip).
#include "gemm_epilogue.h"
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) * lptr_t;

__device__ long long g_big_probe[4];  // wall ticks (100 MHz) of one workgroup: {prologue, K loop, epilogue, shader clocks of the K loop}
void read_big_probe(long long out[4]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_big_probe), 4 * sizeof(long long)); }

// DBG (timing experiments, tools/gemm_bench.py 38-40): 1 = no DMA pieces, 2 = no MFMA, 3 = no barrier/vmcnt wait
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(256, 1) void gemm_big_kernel(GemmP p) {
    constexpr int BM = 256, BN = 256, ROWB = 64, STAGE = (BM + BN) * ROWB, NSLOT = 4;  // 32 KiB per stage
    __shared__ __attribute__((aligned(1024))) char smem[NSLOT * STAGE];
    const long long t0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int row0 = (bid / ntn) * BM, col0 = (bid % ntn) * BN;
    const int lda_b = p.lda * 2, ldw_b = p.ldw * 2;
    const int nkt = p.K / 32;
    const long long a_rows = p.amap.rpg ? ((p.M + p.amap.rpg - 1) / p.amap.rpg) * (long long)p.amap.gstride + p.amap.off + p.amap.rpg : p.M;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)(a_rows * lda_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((long long)p.N * ldw_b), 0x00020000);

    // DMA pieces: piece I = wid + 4 i (i < 4) of each operand = tile rows 16 I .. 16 I + 15, 4 lanes per 64-byte row
    int a_vo[4], w_vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 16 * (wid + 4 * i) + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        int gr = row0 + r;
        if (gr >= p.M) gr = p.M - 1;
        a_vo[i] = ge_map_row(p.amap, gr) * lda_b + q * 16;
        w_vo[i] = (col0 + r) * ldw_b + q * 16;
    }
    const int wave_dst = __builtin_amdgcn_readfirstlane(wid) * 1024;
    (void)a_rs, (void)w_rs, (void)wave_dst;  // (only the device pass uses them: the DMA builtin is hidden from the host pass)
    // one DMA piece: i < 4 -> A piece i of this wave, i >= 4 -> W piece i - 4.  src_st is the (clamped) stage index the
    // bytes come from, st the stage whose slot they land in (they differ only past the end of K, see the loop)
    auto piece = [&](int st, int src_st, int i) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc does not know this builtin)
        if (DBG == 1) return;
        char* base = smem + (st & 3) * STAGE + wave_dst;
        const int ko = src_st * ROWB;
        if (i < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lptr_t)(base + i * 4096), 16, a_vo[i], ko, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lptr_t)(base + BM * ROWB + (i - 4) * 4096), 16, w_vo[i - 4], ko, 0, 0);
#endif
    };
    auto issue = [&](int st, int src_st) {
#pragma unroll
        for (int i = 0; i < 8; ++i) piece(st, src_st, i);
    };

    if constexpr (DBG == 5 || DBG == 6) {
        // DMA stream only, same bytes, two piece geometries: 5 = 8 rows x 128 B (whole cache lines, 64-deep stages, two
        // 64-KiB slots), 6 = 16 rows x 64 B (this kernel's: half a line per row per stage, 32-deep stages, four slots)
#if defined(__HIP_DEVICE_COMPILE__)
        const long long t1 = wall_clock64(), c1 = clock64();
        if constexpr (DBG == 5) {
            int a8[8], w8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 8 * (wid + 4 * i) + (lane >> 3);
                const int q = (lane & 7) ^ ((r >> 1) & 7);
                int gr = row0 + r;
                if (gr >= p.M) gr = p.M - 1;
                a8[i] = gr * lda_b + q * 16;
                w8[i] = (col0 + r) * ldw_b + q * 16;
            }
            for (int st = 0; st < nkt / 2; ++st) {
                char* base = smem + (st & 1) * 65536 + wave_dst;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lptr_t)(base + i * 4096), 16, a8[i], st * 128, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lptr_t)(base + 32768 + i * 4096), 16, w8[i], st * 128, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            }
        } else {
            for (int st = 0; st < nkt; ++st) {
                issue(st, st);
                asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (blockIdx.x == 8 && tid == 0) {
            g_big_probe[0] = 0;
            g_big_probe[1] = wall_clock64() - t1;
            g_big_probe[2] = 0;
            g_big_probe[3] = clock64() - c1;
        }
#endif
        return;
    }

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int sw = (l31 >> 2) & 3;
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = l31 * ROWB + (((2 * s + lh) ^ sw) * 16);
    const int fragA = wr * 128 * ROWB, fragW = BM * ROWB + wc * 128 * ROWB;

    u32x4 fa[2][4], fw[2][4];
    // one fragment read: i < 4 -> A row tile i, i >= 4 -> W row tile i - 4
    auto frag = [&](int set, int st, int s, int i) {
        const char* slot = smem + (st & 3) * STAGE;
        if (i < 4)
            fa[set][i] = *(const u32x4*)(slot + fragA + i * 32 * ROWB + foff[s]);
        else
            fw[set][i - 4] = *(const u32x4*)(slot + fragW + (i - 4) * 32 * ROWB + foff[s]);
    };
    auto frags = [&](int set, int st, int s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) frag(set, st, s, i);
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (DBG == 2) {
                    if (i == j) acc[i][j][0] += __builtin_bit_cast(float, fa[set][i][0] ^ fw[set][j][1]);
                } else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][i]),
                                                                    __builtin_bit_cast(bf16x8, fw[set][j]), acc[i][j], 0, 0, 0);
    };

    // Residual epilogue (out = acc + bias + res, fp32): with one workgroup per CU nothing overlaps the epilogue, so its
    // loads are batched by hand -- 64 per wave in flight (one 32-row strip of the wave tile), strip i+1 fetched while strip i
    // is added and stored (holding strip 0 across the K loop spills: 256 of the 512 registers are accumulators).
    constexpr bool RESEPI = EPI == (GE_RES | GE_F32OUT);
    const bool fastepi = RESEPI && p.cmap.rpg == 0 && row0 + BM <= p.M;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int rbase = row0 + (wu >> 1) * 128, cbase = col0 + (wu & 1) * 128;
    const __amdgpu_buffer_rsrc_t crs =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.Cf, 0, ge_clamp_bytes((long long)p.M * p.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, ge_clamp_bytes((long long)p.M * p.ldr * 4), 0x00020000);
    const int vo_c = (4 * lh * p.ldc + l31) * 4, vo_r = (4 * lh * p.ldr + l31) * 4;
    float R0[64], R1[64];
    auto load_strip = [&](float (&R)[64], int i) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int so_r = ((rbase + i * 32 + (reg & 3) + 8 * (reg >> 2)) * p.ldr + cbase) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                R[reg * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, vo_r + j * 128, so_r, 0));
        }
    };
    // prologue: stages 0..2 in flight, stage 0 landed, its first fragments read  (nkt >= 32: the launcher wants K >= 1024)
    issue(0, 0);
    issue(1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece(2, 2, i);  // (the second half of stage 2 is issued by iteration 0)
    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    frags(0, 0, 0);
    const long long t1 = wall_clock64(), c1 = clock64();
    for (int kt = 0; kt < nkt; ++kt) {
        // k-step 0: second half of stage kt+2's pieces (slot (kt-2)%4, free since the previous barrier)
        const int src2 = kt + 2 < nkt ? kt + 2 : nkt - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            frag(1, kt, 1, i);
            if (i & 1) piece(kt + 2, src2, 4 + (i >> 1));
        }
        mma(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read (DMA piece)
        }
        __builtin_amdgcn_sched_barrier(0);
        // k-step 1
        // (issues and reads past the last stage are made anyway -- clamped source, free slot -- so that the loop body is
        // ONE basic block the scheduler can interleave and the vmcnt immediate is a constant)
        if (DBG != 3) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        // (reads and DMA pieces alternate in SOURCE order: the compiler must assume they alias, so it keeps that order)
        const int src3 = kt + 3 < nkt ? kt + 3 : nkt - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            frag(0, kt + 1, 0, i);
            if (i & 1) piece(kt + 3, src3, i >> 1);  // first half of stage kt+3's pieces (slot (kt-1)%4)
        }
        mma(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read (DMA piece)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const long long t2 = wall_clock64(), c2 = clock64();
    bool done = false;
    if constexpr (RESEPI) {
        if (fastepi) {
            float bj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bj[j] = p.bias ? p.bias[cbase + j * 32 + l31] : 0.f;
            auto put_strip = [&](const float (&R)[64], int i) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int so_c = ((rbase + i * 32 + (reg & 3) + 8 * (reg >> 2)) * p.ldc + cbase) * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = acc[i][j][reg] + bj[j] + R[reg * 4 + j];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, vo_c + j * 128, so_c, 0);
                    }
                }
            };
            load_strip(R0, 0);
            load_strip(R1, 1);
            put_strip(R0, 0);
            load_strip(R0, 2);
            put_strip(R1, 1);
            load_strip(R1, 3);
            put_strip(R0, 2);
            put_strip(R1, 3);
            done = true;
        }
    }
    if (!done) gemm_epilogue<EPI, 4, 4>(p, acc, rbase, cbase, row0, BM, lane);
    if (p.variant >= 37 && blockIdx.x == 8 && tid == 0) {
        g_big_probe[0] = t1 - t0;
        g_big_probe[1] = t2 - t1;
        g_big_probe[2] = wall_clock64() - t2;
        g_big_probe[3] = c2 - c1;
    }
}

// returns false when the shape / epilogue is not covered
bool launch_gemm_big(const GemmP& p, hipStream_t st) {
    if (p.K % 32 != 0 || p.K < 1024 || p.N % 256 != 0) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 8) || (p.ldw % 8)) return false;
    const long long a_rows = p.amap.rpg ? ((p.M + p.amap.rpg - 1) / p.amap.rpg) * (long long)p.amap.gstride + p.amap.off + p.amap.rpg : p.M;
    if (a_rows * p.lda * 2 >= 0x7ffff000ll || (long long)p.N * p.ldw * 2 >= 0x7ffff000ll) return false;
    const bool f32out = p.Cf != nullptr;
    const int epi = (p.gelu ? GE_GELU : 0) | (p.res ? GE_RES : 0) | (p.rowtab ? GE_ROWTAB : 0) | (f32out ? GE_F32OUT : 0);
    const dim3 grid(((p.M + 255) / 256) * (p.N / 256)), block(256);
    switch (epi) {
        case 0: hipLaunchKernelGGL(gemm_big_kernel<0>, grid, block, 0, st, p); return true;
        case GE_F32OUT: hipLaunchKernelGGL(gemm_big_kernel<GE_F32OUT>, grid, block, 0, st, p); return true;
        case GE_GELU: hipLaunchKernelGGL(gemm_big_kernel<GE_GELU>, grid, block, 0, st, p); return true;
        case GE_GELU | GE_F32OUT: hipLaunchKernelGGL((gemm_big_kernel<GE_GELU | GE_F32OUT>), grid, block, 0, st, p); return true;
        case GE_RES | GE_F32OUT:
            if (p.variant == 38) hipLaunchKernelGGL((gemm_big_kernel<GE_RES | GE_F32OUT, 1>), grid, block, 0, st, p);
            else if (p.variant == 39) hipLaunchKernelGGL((gemm_big_kernel<GE_RES | GE_F32OUT, 2>), grid, block, 0, st, p);
            else if (p.variant == 40) hipLaunchKernelGGL((gemm_big_kernel<GE_RES | GE_F32OUT, 3>), grid, block, 0, st, p);
            else if (p.variant == 41) hipLaunchKernelGGL((gemm_big_kernel<GE_RES | GE_F32OUT, 5>), grid, block, 0, st, p);
            else if (p.variant == 42) hipLaunchKernelGGL((gemm_big_kernel<GE_RES | GE_F32OUT, 6>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((gemm_big_kernel<GE_RES | GE_F32OUT, 0>), grid, block, 0, st, p);
            return true;
        default: return false;
    }
}

}  // namespace m3pc
