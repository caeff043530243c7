// bf16 MFMA GEMM, 128x128 tile, operands staged THROUGH REGISTERS (experimental, GemmP::variant 36).
//
// The ring kernel (gemm_glds.hip) is bound by the issue of its LDS-DMA pieces (issue stalls ~50 % of wave
// cycles, <1 % of them LDS).  This variant moves the same bytes with plain 16-byte buffer loads into VGPRs and
// ds_write_b128 into a two-slot LDS image -- the classic path -- with the ring kernel's other ingredients kept: 32-deep
// stages, two stages in flight (two register sets), one barrier per stage, the same XOR-swizzled 64-byte LDS rows
// (applied on the ds_write side here), scalar-addressed epilogue, 32 KiB of LDS per workgroup.
//   iteration kt:  barrier (stage kt visible) | issue loads of stage kt+2 -> set (kt&1) | MFMAs of stage kt out of
//                  slot kt&1 | park stage kt+1 (set (kt+1)&1, loaded during iteration kt-1) in slot (kt+1)&1
#include "gemm_epilogue.h"
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(256, 4) void gemm_rs_kernel(GemmP p) {
    constexpr int BM = 128, BN = 128, ROWB = 64, BUF = (BM + BN) * ROWB;  // 16 KiB per stage
    __shared__ __attribute__((aligned(1024))) char smem[2 * BUF];
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

    // piece I = wid + 4 i (i < 2) of each operand: 16 rows x 64 B, lane -> (row 16 I + lane/4, chunk lane%4)
    int a_vo[2], w_vo[2], dstA[2], dstW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * (wid + 4 * i) + lane / 4, q = lane % 4;
        int gr = row0 + r;
        if (gr >= p.M) gr = p.M - 1;
        a_vo[i] = ge_map_row(p.amap, gr) * lda_b + q * 16;
        w_vo[i] = (col0 + r) * ldw_b + q * 16;
        const int sw = (q ^ ((r >> 2) & 3)) * 16;  // LDS-side swizzle: logical chunk q of row r sits at q ^ ((r>>2)&3)
        dstA[i] = r * ROWB + sw;
        dstW[i] = BM * ROWB + r * ROWB + sw;
    }
    struct Stage {
        u32x4 a[2], w[2];
    };
    auto gload = [&](Stage& s, int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            s.a[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[i], kt * ROWB, 0);
            s.w[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_vo[i], kt * ROWB, 0);
        }
    };
    auto park = [&](const Stage& s, int slot) {
        char* b = smem + slot * BUF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(u32x4*)(b + dstA[i]) = s.a[i];
            *(u32x4*)(b + dstW[i]) = s.w[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int sw = (l31 >> 2) & 3;
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = l31 * ROWB + (((2 * s + lh) ^ sw) * 16);
    const int fragA = wr * 64 * ROWB, fragW = BM * ROWB + wc * 64 * ROWB;

    Stage s0, s1;
    gload(s0, 0);
    gload(s1, 1);
    park(s0, 0);
    auto body = [&](int kt, Stage& nxt2, const Stage& nxt1) {
        // nxt1 holds stage kt+1 (loaded during iteration kt-1); nxt2 (the set stage kt just vacated) takes stage kt+2
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        gload(nxt2, kt + 2 < nkt ? kt + 2 : nkt - 1);  // unconditional (clamped) so the compiler can COUNT vmcnt
        __builtin_amdgcn_sched_barrier(0);
        const char* cur = smem + (kt & 1) * BUF;
        u32x4 fa[2][2], fw[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[s][i] = *(const u32x4*)(cur + fragA + i * 32 * ROWB + foff[s]);
                fw[s][i] = *(const u32x4*)(cur + fragW + i * 32 * ROWB + foff[s]);
            }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s][i]),
                                                                        __builtin_bit_cast(bf16x8, fw[s][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        park(nxt1, (kt + 1) & 1);  // past the end this parks a stage nobody reads
    };
    // register sets alternate: stage kt lives in s(kt&1) until parked during iteration kt-1
    for (int kt = 0; kt < nkt; kt += 2) {
        body(kt, s0, s1);      // stage kt+2 -> s0 (stage kt was parked from s0 already), park stage kt+1 from s1
        body(kt + 1, s1, s0);  // stage kt+3 -> s1, park stage kt+2 from s0
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    gemm_epilogue<EPI, 2, 2>(p, acc, row0 + (wu >> 1) * 64, col0 + (wu & 1) * 64, row0, BM, lane);
}

// returns false when the shape / epilogue is not covered
bool launch_gemm_rs(const GemmP& p, hipStream_t st) {
    if (p.K % 64 != 0 || p.N % 128 != 0 || p.K < 128) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 8) || (p.ldw % 8)) return false;
    const long long a_rows = p.amap.rpg ? ((p.M + p.amap.rpg - 1) / p.amap.rpg) * (long long)p.amap.gstride + p.amap.off + p.amap.rpg : p.M;
    if (a_rows * p.lda * 2 >= 0x7ffff000ll || (long long)p.N * p.ldw * 2 >= 0x7ffff000ll) return false;
    const bool f32out = p.Cf != nullptr;
    const int epi = (p.gelu ? GE_GELU : 0) | (p.res ? GE_RES : 0) | (p.rowtab ? GE_ROWTAB : 0) | (f32out ? GE_F32OUT : 0);
    const dim3 grid(((p.M + 127) / 128) * (p.N / 128)), block(256);
    switch (epi) {
        case 0: hipLaunchKernelGGL(gemm_rs_kernel<0>, grid, block, 0, st, p); return true;
        case GE_F32OUT: hipLaunchKernelGGL(gemm_rs_kernel<GE_F32OUT>, grid, block, 0, st, p); return true;
        case GE_GELU: hipLaunchKernelGGL(gemm_rs_kernel<GE_GELU>, grid, block, 0, st, p); return true;
        case GE_GELU | GE_F32OUT: hipLaunchKernelGGL(gemm_rs_kernel<GE_GELU | GE_F32OUT>, grid, block, 0, st, p); return true;
        case GE_RES | GE_F32OUT: hipLaunchKernelGGL(gemm_rs_kernel<GE_RES | GE_F32OUT>, grid, block, 0, st, p); return true;
        case GE_ROWTAB | GE_F32OUT: hipLaunchKernelGGL(gemm_rs_kernel<GE_ROWTAB | GE_F32OUT>, grid, block, 0, st, p); return true;
        default: return false;
    }
}

}  // namespace m3pc
