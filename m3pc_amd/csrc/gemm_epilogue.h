// Shared epilogue of the bf16 MFMA GEMM kernels (gemm_glds.hip, gemm_persist.hip).
//
// Accumulator layout of v_mfma_f32_32x32x16_bf16: acc[i][j][reg] is row (reg&3) + 8*(reg>>2) + 4*(lane>>5),
// column lane&31 of 32x32 tile (i, j) of the wave tile.
//
// Fast path (full tile, identity row map): addresses are split into a wave-uniform part that lives in SGPRs
// (buffer soffset = row * ld) and ONE per-lane 32-bit offset computed once (buffer voffset); the j-tile
// distance is an immediate.  Per element that leaves: bias add, [GELU], [residual add], convert, store --
// no per-element 64-bit address arithmetic (hipcc otherwise spends ~2 VALU instructions per element on it,
// which at K = 512 costs as much as a third of the tile's MFMA time).
#pragma once
#include "kernels.h"

namespace m3pc {

enum { GE_GELU = 1, GE_RES = 2, GE_ROWTAB = 4, GE_F32OUT = 8 };

typedef float ge_f32x16 __attribute__((ext_vector_type(16)));

// exact-erf GELU via Abramowitz-Stegun 7.1.26 (|err(erf)| <= 1.5e-7): 2 transcendentals + 8 VALU
__device__ __forceinline__ float ge_gelu(float x) {
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

// the same arithmetic on two values at once: the polynomial runs on v_pk_fma_f32 / v_pk_mul_f32 (two fp32 lanes per
// instruction), which halves its VALU cost; every operation and its order match ge_gelu, so the results are identical
typedef float ge_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ge_f32x2 ge_gelu2(ge_f32x2 x) {
    const ge_f32x2 one = {1.0f, 1.0f};
    const ge_f32x2 ax = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const ge_f32x2 d = __builtin_elementwise_fma((ge_f32x2){0.3275911f, 0.3275911f}, ax, one);
    const ge_f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    ge_f32x2 p = __builtin_elementwise_fma(t, (ge_f32x2){1.061405429f, 1.061405429f}, (ge_f32x2){-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(t, p, (ge_f32x2){1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(t, p, (ge_f32x2){-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(t, p, (ge_f32x2){0.254829592f, 0.254829592f});
    p *= t;
    const ge_f32x2 a = -ax * ax * 1.44269504088896340736f;
    const ge_f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const ge_f32x2 erf_abs = __builtin_elementwise_fma(-p, e, one);
    const ge_f32x2 hx = 0.5f * x;
    return __builtin_elementwise_fma(__builtin_elementwise_abs(hx), erf_abs, hx);
}

__device__ __forceinline__ int ge_map_row(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}

__device__ __forceinline__ unsigned ge_clamp_bytes(long long b) { return b > 0xfffff000ll ? 0xfffff000u : (unsigned)b; }

// rbase / cbase: first row / column of this WAVE's tile, wave-uniform (pass values derived from readfirstlane)
// bias_pre: the TN bias values of this lane's columns when the caller fetched them earlier (a persistent kernel has
// the next tile's DMA in flight here and a load issued now would wait for all of it), else nullptr
template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, ge_f32x16 (&acc)[TM][TN], int rbase, int cbase, int tile_row0,
                                              int tile_rows, int lane, const float* bias_pre = nullptr) {
    const int l31 = lane & 31, lh = lane >> 5;
    float bj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bj[j] = bias_pre ? bias_pre[j] : (p.bias ? p.bias[cbase + j * 32 + l31] : 0.f);
    const bool fast = p.cmap.rpg == 0 && tile_row0 + tile_rows <= p.M && !(EPI & GE_ROWTAB);
    if (fast) {
        constexpr int ES = (EPI & GE_F32OUT) ? 4 : 2;
        void* cptr = (EPI & GE_F32OUT) ? (void*)p.Cf : (void*)p.Cb;
        const __amdgpu_buffer_rsrc_t crs =
            __builtin_amdgcn_make_buffer_rsrc(cptr, 0, ge_clamp_bytes((long long)p.M * p.ldc * ES), 0x00020000);
        const int vo_c = (4 * lh * p.ldc + l31) * ES;
        __amdgpu_buffer_rsrc_t rrs = crs;
        int vo_r = 0;
        if constexpr (EPI & GE_RES) {
            rrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, ge_clamp_bytes((long long)p.M * p.ldr * 4), 0x00020000);
            vo_r = (4 * lh * p.ldr + l31) * 4;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rr = rbase + i * 32 + (reg & 3) + 8 * (reg >> 2);
                const int so_c = (rr * p.ldc + cbase) * ES;
                const int so_r = (rr * p.ldr + cbase) * 4;
                float vj[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) vj[j] = acc[i][j][reg] + bj[j];
                if constexpr ((EPI & GE_GELU) != 0) {
                    if constexpr (TN % 2 == 0) {
#pragma unroll
                        for (int j = 0; j < TN; j += 2) {
                            const ge_f32x2 g2 = ge_gelu2((ge_f32x2){vj[j], vj[j + 1]});
                            vj[j] = g2.x;
                            vj[j + 1] = g2.y;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < TN; ++j) vj[j] = ge_gelu(vj[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v = vj[j];
                    if constexpr (EPI & GE_RES)
                        v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, vo_r + j * 128, so_r, 0));
                    if constexpr (EPI & GE_F32OUT)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, vo_c + j * 128, so_c, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)v), crs, vo_c + j * 64,
                                                              so_c, 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = rbase + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (r < p.M) {
                const long long pr = ge_map_row(p.cmap, r);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int c = cbase + j * 32 + l31;
                    float v = acc[i][j][reg] + bj[j];
                    if constexpr (EPI & GE_ROWTAB) v += p.rowtab[(long long)(r % p.rt_mod) * p.rt_ld + c];
                    if constexpr (EPI & GE_GELU) v = ge_gelu(v);
                    if constexpr (EPI & GE_RES) v += p.res[pr * p.ldr + c];
                    if constexpr (EPI & GE_F32OUT)
                        p.Cf[pr * p.ldc + c] = v;
                    else
                        p.Cb[pr * p.ldc + c] = (bf16_t)v;
                }
            }
        }
    }
}

}  // namespace m3pc
