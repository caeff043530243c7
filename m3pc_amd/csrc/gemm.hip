// MFMA GEMM for gfx950: C = epilogue(A * W^T), A [M][K] activations, W [N][K] weights (torch Linear
// layout, K contiguous => both MFMA operands are read as contiguous 16-byte K fragments).
//
// One kernel body serves both operand types through a shared "128 bytes of K per LDS row" tile:
//   bf16: 64 k per row; quarter s (32 B) = one v_mfma_f32_32x32x16_bf16, lane half h reads the 16 B at
//         32s+16h (= k 16s+8h .. +7), exactly the instruction's A/B lane map.
//   fp32: 32 k per row; the same 16 B hold 4 floats k = 8s+4h+e; they feed four
//         v_mfma_f32_32x32x2_f32 (e = 0..3), each pairing k = 8s+e (h=0) with 8s+4+e (h=1).  A and B use
//         the same permutation, so the dot product is complete and each product is an exact fp32 fma.
// LDS rows are padded to 144 B: the 16-lane groups of ds_read_b128 then hit 16 distinct 4-bank slots.
//
// Pipeline: two LDS buffers; tile k+1 travels HBM/L2 -> registers while tile k is multiplied, and is
// written to the other buffer after the MFMAs: one barrier per k-tile.
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define LDS_ROW 144

// epilogue flags (template parameter)
enum { EPI_GELU = 1, EPI_RES = 2, EPI_ROWTAB = 4, EPI_F32OUT = 8, EPI_SPLITK = 16 };

// exact-erf GELU.  fp32 path: libm erff.  bf16 path: Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 on erf,
// far below bf16 resolution) -- 2 transcendentals instead of a ~35-instruction erff, which would
// otherwise make the FFN epilogue VALU-bound beside the matrix pipe.
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_fast(float x) {
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
    return fmaf(fabsf(hx), erf_abs, hx);  // 0.5x(1+erf(x/sqrt2)) with erf odd
}

__device__ __forceinline__ int map_row(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}

template <typename T, int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    constexpr int TM = BM / 64, TN = BN / 64;                // 32x32 MFMA tiles per wave (2x2 waves)
    constexpr int A_CH = BM * 8 / 256, W_CH = BN * 8 / 256;  // 16-byte chunks per thread per k-tile
    constexpr int BUF = (BM + BN) * LDS_ROW;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles (column tiles of one row panel share A through that XCD's L2).
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

    const char* Ab = (const char*)p.A;
    const char* Wb = (const char*)p.W;
    const long long lda_b = (long long)p.lda * sizeof(T), ldw_b = (long long)p.ldw * sizeof(T);
    const int nkt = (int)((long long)p.K * sizeof(T) / 128);
    int kt0 = 0, kt1 = nkt;  // split-K: blockIdx.y owns a contiguous run of k-tiles, raw partials go to p.ws
    if constexpr (EPI & EPI_SPLITK) {
        kt0 = (int)((long long)nkt * blockIdx.y / gridDim.y);
        kt1 = (int)((long long)nkt * (blockIdx.y + 1) / gridDim.y);
    }

    // per-thread global sources (fixed rows, advancing 128 B per k-tile) and LDS destinations
    const char* a_src[A_CH];
    int a_dst[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + i * 256, r = c >> 3, kc = c & 7;
        int gr = row0 + r;
        if (gr >= p.M) gr = p.M - 1;
        a_src[i] = Ab + (long long)map_row(p.amap, gr) * lda_b + kc * 16 + (long long)kt0 * 128;
        a_dst[i] = r * LDS_ROW + kc * 16;
    }
    const char* w_src[W_CH];
    int w_dst[W_CH];
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int c = tid + i * 256, r = c >> 3, kc = c & 7;
        w_src[i] = Wb + (long long)(col0 + r) * ldw_b + kc * 16 + (long long)kt0 * 128;
        w_dst[i] = BM * LDS_ROW + r * LDS_ROW + kc * 16;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Register prefetch two k-tiles deep: while tile k is multiplied out of LDS, tiles k+1 and k+2 are
    // in flight from L2/HBM in two register sets (few-row GEMMs run one block per CU, so nothing else
    // hides the load latency of their short K chains).
    const int nloc = kt1 - kt0;
    u32x4 ra0[A_CH], rw0[W_CH], ra1[A_CH], rw1[W_CH];
    auto gload = [&](u32x4* ra, u32x4* rw, int kt) {
        const long long ko = (long long)kt * 128;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i] = *(const u32x4*)(a_src[i] + ko);
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[i] = *(const u32x4*)(w_src[i] + ko);
    };
    auto lstore = [&](const u32x4* ra, const u32x4* rw, char* buf) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *(u32x4*)(buf + a_dst[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < W_CH; ++i) *(u32x4*)(buf + w_dst[i]) = rw[i];
    };
    gload(ra0, rw0, 0);
    lstore(ra0, rw0, smem);
    if (nloc > 1) gload(ra0, rw0, 1);
    if (nloc > 2) gload(ra1, rw1, 2);
    __syncthreads();

    const int fragA = (wr * (BM / 2) + (lane & 31)) * LDS_ROW + 16 * (lane >> 5);
    const int fragW = BM * LDS_ROW + (wc * (BN / 2) + (lane & 31)) * LDS_ROW + 16 * (lane >> 5);

    auto compute = [&](const char* cur) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const u32x4*)(cur + fragA + i * 32 * LDS_ROW + 32 * s);
#pragma unroll
            for (int j = 0; j < TN; ++j) fw[j] = *(const u32x4*)(cur + fragW + j * 32 * LDS_ROW + 32 * s);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fw[j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 a4 = __builtin_bit_cast(f32x4, fa[i]);
                        const f32x4 w4 = __builtin_bit_cast(f32x4, fw[j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], w4[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
    // step kt: multiply tile kt; park tile kt+1 (held in `ra`) in the other LDS buffer; refill `ra` with kt+3
    auto step = [&](u32x4* ra, u32x4* rw, int kt) {
        compute(smem + (kt & 1) * BUF);
        if (kt + 1 < nloc) lstore(ra, rw, smem + ((kt + 1) & 1) * BUF);
        __syncthreads();
        if (kt + 3 < nloc) gload(ra, rw, kt + 3);
    };
    for (int kt = 0; kt < nloc; kt += 2) {
        step(ra0, rw0, kt);
        if (kt + 1 < nloc) step(ra1, rw1, kt + 1);
    }

    // epilogue.  acc[i][j][reg]: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31 of the 32x32 tile
    if constexpr (EPI & EPI_SPLITK) {
        float* slab = p.ws + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = row0 + wr * (BM / 2) + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (r < p.M) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        slab[(long long)r * p.N + col0 + wc * (BN / 2) + j * 32 + (lane & 31)] = acc[i][j][reg];
                }
            }
        return;
    }
    float bj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bj[j] = p.bias ? p.bias[col0 + wc * (BN / 2) + j * 32 + (lane & 31)] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = row0 + wr * (BM / 2) + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (r < p.M) {
                const long long pr = map_row(p.cmap, r);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int c = col0 + wc * (BN / 2) + j * 32 + (lane & 31);
                    float v = acc[i][j][reg] + bj[j];
                    if constexpr (EPI & EPI_ROWTAB) v += p.rowtab[(long long)(r % p.rt_mod) * p.rt_ld + c];
                    if constexpr (EPI & EPI_GELU) v = sizeof(T) == 2 ? gelu_fast(v) : gelu_exact(v);
                    if constexpr (EPI & EPI_RES) v += p.res[pr * p.ldr + c];
                    if constexpr ((EPI & EPI_F32OUT) || sizeof(T) == 4)
                        p.Cf[pr * p.ldc + c] = v;
                    else
                        p.Cb[pr * p.ldc + c] = (bf16_t)v;
                }
            }
        }
    }
}

// second half of a split-K GEMM: sum the S raw slabs in a fixed order (deterministic) and apply the epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p, int S, int exact_gelu) {
    const long long n = (long long)p.M * p.N;
    for (long long x = blockIdx.x * 256LL + threadIdx.x; x < n; x += (long long)gridDim.x * 256) {
        const int r = (int)(x / p.N), c = (int)(x % p.N);
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += p.ws[(long long)s * n + x];
        if (p.bias) v += p.bias[c];
        if (p.rowtab) v += p.rowtab[(long long)(r % p.rt_mod) * p.rt_ld + c];
        if (p.gelu) v = exact_gelu ? gelu_exact(v) : gelu_fast(v);
        const long long pr = map_row(p.cmap, r);
        if (p.res) v += p.res[pr * p.ldr + c];
        if (p.Cf) p.Cf[pr * p.ldc + c] = v;
        if (p.Cb) p.Cb[pr * p.ldc + c] = (bf16_t)v;
    }
}

// Same reduction, one wave per output row, followed by the LayerNorm that consumes the row (GemmP::ln_*): saves the
// separate LayerNorm launch of the few-row fp32 passes.  Column layout and summation order are those of
// layernorm_vec_kernel (elementwise.hip), so fused and unfused paths give identical bits.
typedef float f32x4r __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float wave_sum_r(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(GemmP p, int S, int exact_gelu) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.M) return;
    const long long mn = (long long)p.M * p.N;
    const int n = p.N >> 8;  // slabs of 256 columns (<= 4)
    f32x4r v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
            const int c = i * 256 + lane * 4;
            f32x4r a = {0.f, 0.f, 0.f, 0.f};
            const float* src = p.ws + (long long)r * p.N + c;
            int k = 0;
            for (; k + 4 <= S; k += 4) {  // four slabs in flight, added in slab order (the sum order is part of the result)
                const f32x4r t0 = *(const f32x4r*)(src + (k + 0) * mn), t1 = *(const f32x4r*)(src + (k + 1) * mn);
                const f32x4r t2 = *(const f32x4r*)(src + (k + 2) * mn), t3 = *(const f32x4r*)(src + (k + 3) * mn);
                a += t0;
                a += t1;
                a += t2;
                a += t3;
            }
            for (; k < S; ++k) a += *(const f32x4r*)(src + k * mn);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = a[e];
                if (p.bias) x += p.bias[c + e];
                if (p.rowtab) x += p.rowtab[(long long)(r % p.rt_mod) * p.rt_ld + c + e];
                if (p.gelu) x = exact_gelu ? gelu_exact(x) : gelu_fast(x);
                if (p.res) x += p.res[(long long)r * p.ldr + c + e];
                a[e] = x;
            }
            *(f32x4r*)(p.Cf + (long long)r * p.ldc + c) = a;
            v[i] = a;
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
    const float inv_d = 1.0f / (float)p.N;
    const float mean = wave_sum_r(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = v[i][e] - mean;
                q += c * c;
            }
        }
    const float rstd = rsqrtf(wave_sum_r(q) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
            const int c = i * 256 + lane * 4;
            const f32x4r g = *(const f32x4r*)(p.ln_g + c);
            const f32x4r b = *(const f32x4r*)(p.ln_b + c);
            f32x4r y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            *(f32x4r*)(p.ln_out + (long long)r * p.N + c) = y;
        }
}

template <typename T, int BM, int BN, int EPI>
static void launch_cfg(const GemmP& p, hipStream_t st) {
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI>), dim3(grid), dim3(256), 0, st, p);
}

template <typename T, int EPI>
static int launch_epi(const GemmP& p, hipStream_t st) {
    // 128x128 tiles only when they still fill the chip twice over; otherwise 64x64 tiles (4x the blocks):
    // the batch-1 policy pass and the top-k re-score are latency-bound on few rows.
    const long long big_tiles = (long long)((p.M + 127) / 128) * (p.N / 128);
    if ((p.N % 128) != 0 || big_tiles < 512) {
        // few tiles and a long serial K chain per tile (fp32: 4 f32-MFMAs per 16 bytes of K): split K
        // over blocks into raw slabs, reduce + epilogue in a second small kernel.
        const long long tiles = (long long)((p.M + 63) / 64) * (p.N / 64);
        const int nkt = (int)((long long)p.K * sizeof(T) / 128);
        int S = 1;
        // measured (tools/gemm_bench_f32.py): with >= 200 tiles and a 16-k-tile chain (K = 512) one pass beats
        // split + reduce (top-16 re-score: QKV 23.9 vs 31.3 us, FFN1 27.8 vs 38.5 us); K = 2048 chains and the
        // batch-1 policy pass (2-32 tiles) still want the split
        if (sizeof(T) == 4 && p.ws && tiles < 768 && (tiles < 200 || nkt > 16)) {
            S = (int)((1023 + tiles) / tiles);
            if (S > nkt / 4) S = nkt / 4;
            if (S > 16) S = 16;
            while (S > 1 && (long long)S * p.M * p.N * 4 > p.ws_bytes) --S;
        }
        if (S > 1) {
            hipLaunchKernelGGL((gemm_kernel<T, 64, 64, EPI_SPLITK>), dim3((unsigned)tiles, S), dim3(256), 0, st, p);
            if (p.ln_g && p.ln_out && p.Cf && !p.Cb && p.cmap.rpg == 0 && p.N % 256 == 0 && p.N <= 1024 && p.ldc % 4 == 0 &&
                (!p.res || p.ldr % 4 == 0)) {
                hipLaunchKernelGGL(splitk_reduce_ln_kernel, dim3((p.M + 3) / 4), dim3(256), 0, st, p, S, (int)(sizeof(T) == 4));
                return 1;
            }
            const long long n = (long long)p.M * p.N;
            const int g = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, st, p, S, (int)(sizeof(T) == 4));
        } else {
            launch_cfg<T, 64, 64, EPI>(p, st);
        }
    } else {
        launch_cfg<T, 128, 128, EPI>(p, st);
    }
    return 0;
}

template <typename T>
static int launch_t(const GemmP& p, hipStream_t st) {
    const bool f32out = sizeof(T) == 4 || p.Cf != nullptr;
    const int epi = (p.gelu ? EPI_GELU : 0) | (p.res ? EPI_RES : 0) | (p.rowtab ? EPI_ROWTAB : 0) |
                    (f32out && sizeof(T) == 2 ? EPI_F32OUT : 0);
    switch (epi) {
        case 0: return launch_epi<T, 0>(p, st);
        case EPI_F32OUT: return launch_epi<T, EPI_F32OUT>(p, st);
        case EPI_GELU: return launch_epi<T, EPI_GELU>(p, st);
        case EPI_GELU | EPI_F32OUT: return launch_epi<T, EPI_GELU | EPI_F32OUT>(p, st);
        case EPI_RES: return launch_epi<T, EPI_RES>(p, st);
        case EPI_RES | EPI_F32OUT: return launch_epi<T, EPI_RES | EPI_F32OUT>(p, st);
        case EPI_ROWTAB: return launch_epi<T, EPI_ROWTAB>(p, st);
        case EPI_ROWTAB | EPI_F32OUT: return launch_epi<T, EPI_ROWTAB | EPI_F32OUT>(p, st);
        default: return 0;  // no caller combines the remaining flags
    }
}

// returns 1 when the LayerNorm described by p.ln_* was applied as part of the launch (see GemmP), else 0
int launch_gemm(const GemmP& p, int dtype, hipStream_t st) {
    if (p.M <= 0) return 0;
    // default (variant 0): many-row bf16 problems on the LDS-DMA kernel (gemm_glds.hip, 128x128 tiles),
    // few-row problems on the register-staged kernel below; variants 4-9 are the experimental tilings
#ifdef M3PC_LAB  // experimental tilings kept for tools/gemm_bench.py: compiled into libm3pc_hip_lab.so only
    if (dtype == DT_BF16 && p.variant == 9 && launch_gemm_persist(p, st)) return 0;
    if (dtype == DT_BF16 && p.variant == 36 && launch_gemm_rs(p, st)) return 0;
#endif
    // long-K many-row problems (this step's FFN2, K = 2048): 256x256 tiles at one wave per SIMD (gemm_big.hip).  Same
    // MFMA instruction, k order and epilogue arithmetic as the 128x128 ring kernel, so results are bit-identical and the
    // choice may depend on the row count.  It needs about one tile per CU to pay (one workgroup per CU, no overlap).
    static const bool no_big = M3PC_ENV("M3PC_NO_GEMM_BIG") != nullptr;  // A/B switch
    if (dtype == DT_BF16 && p.variant >= 37 && p.variant <= 42 && launch_gemm_big(p, st)) return 0;
    if (dtype == DT_BF16 && p.variant >= 43 && p.variant <= 46 && launch_gemm_line(p, p.variant == 44 ? 256 : 128, st)) return 0;
    // K = 512-class many-row problems without a residual stream: 128x128 tiles fed by whole-cache-line DMA pieces through
    // a five-unit ring, persistent workgroups (gemm_line.hip).  The residual GEMMs of this class are HBM-bound and stay
    // on the three-slot ring kernel, as do output row maps / row tables.
    static const bool no_line = M3PC_ENV("M3PC_NO_GEMM_LINE") != nullptr;  // A/B switch
    static const long long line_min = M3PC_ENV("M3PC_LINE_MIN_TILES") ? atoll(M3PC_ENV("M3PC_LINE_MIN_TILES")) : 256;  // (one tile per CU at least: the head GEMMs of a candidate half, 256 tiles, take 12-15 us here against 22-26)
    if (dtype == DT_BF16 && p.variant == 0 && !no_line && !p.res && p.K < 1024 &&
        (long long)((p.M + 127) / 128) * (p.N / 128) >= line_min && launch_gemm_line(p, 128, st))
        return 0;
    if (dtype == DT_BF16 && p.variant == 0 && !no_big && p.K >= 1024 && (long long)((p.M + 255) / 256) * (p.N / 256) >= 224 &&
        launch_gemm_big(p, st))
        return 0;
#ifdef M3PC_LAB
    if (dtype == DT_BF16 && p.variant >= 7 && p.variant != 9 && p.variant < 20 && launch_gemm_ring(p, st)) return 0;
#endif
    if (dtype == DT_BF16 && (p.variant == 0 || (p.variant >= 2 && p.variant < 7) || (p.variant >= 20 && p.variant <= 32)) &&
        launch_gemm_glds(p, st))
        return 0;
    if (dtype == DT_BF16) return launch_t<bf16_t>(p, st);
    // few-row fp32 GEMMs: K split inside a 16-wave workgroup, one launch (only where split-K is allowed at all, i.e.
    // where every rank runs the same row count: p.ws is set exactly then).  variant 2 keeps the slab path for A/B runs.
    static const bool no_direct = M3PC_ENV("M3PC_NO_F32_DIRECT") != nullptr;  // A/B switch
    if (p.ws && p.variant != 2 && !no_direct && launch_gemm_f32_direct(p, st)) return 0;
    return launch_t<float>(p, st);
}

}  // namespace m3pc
