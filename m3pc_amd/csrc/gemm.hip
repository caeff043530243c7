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
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define LDS_ROW 144

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ int map_row(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    constexpr int TM = BM / 64, TN = BN / 64;         // 32x32 MFMA tiles per wave (2x2 waves)
    constexpr int A_CH = BM * 8 / 256, W_CH = BN * 8 / 256;  // 16-byte chunks per thread per k-tile
    __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * LDS_ROW];
    char* sA = smem;
    char* sW = smem + BM * LDS_ROW;

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

    // per-thread global source rows (fixed over the K loop)
    const char* a_src[A_CH];
    int a_dst[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + i * 256, r = c >> 3, kc = c & 7;
        int gr = row0 + r;
        if (gr >= p.M) gr = p.M - 1;
        a_src[i] = Ab + (long long)map_row(p.amap, gr) * lda_b + kc * 16;
        a_dst[i] = r * LDS_ROW + kc * 16;
    }
    const char* w_src[W_CH];
    int w_dst[W_CH];
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int c = tid + i * 256, r = c >> 3, kc = c & 7;
        w_src[i] = Wb + (long long)(col0 + r) * ldw_b + kc * 16;
        w_dst[i] = r * LDS_ROW + kc * 16;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 ra[A_CH], rw[W_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ra[i] = *(const uint4*)(a_src[i]);
#pragma unroll
    for (int i = 0; i < W_CH; ++i) rw[i] = *(const uint4*)(w_src[i]);

    const int fragA = (wr * (BM / 2) + (lane & 31)) * LDS_ROW + 16 * (lane >> 5);
    const int fragW = (wc * (BN / 2) + (lane & 31)) * LDS_ROW + 16 * (lane >> 5);

    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *(uint4*)(sA + a_dst[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < W_CH; ++i) *(uint4*)(sW + w_dst[i]) = rw[i];
        __syncthreads();
        if (kt + 1 < nkt) {
            const long long ko = (long long)(kt + 1) * 128;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) ra[i] = *(const uint4*)(a_src[i] + ko);
#pragma unroll
            for (int i = 0; i < W_CH; ++i) rw[i] = *(const uint4*)(w_src[i] + ko);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(sA + fragA + i * 32 * LDS_ROW + 32 * s);
#pragma unroll
            for (int j = 0; j < TN; ++j) fw[j] = *(const uint4*)(sW + fragW + j * 32 * LDS_ROW + 32 * s);
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
        __syncthreads();
    }

    // epilogue.  acc[i][j][reg]: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31 of the 32x32 tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = row0 + wr * (BM / 2) + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (r >= p.M) continue;
            const long long pr = map_row(p.cmap, r);
            const float* rt = p.rowtab ? p.rowtab + (long long)(r % p.rt_mod) * p.rt_ld : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = col0 + wc * (BN / 2) + j * 32 + (lane & 31);
                float v = acc[i][j][reg];
                if (p.bias) v += p.bias[c];
                if (rt) v += rt[c];
                if (p.gelu) v = gelu_erf(v);
                if (p.res) v += p.res[pr * p.ldr + c];
                if (p.Cf) p.Cf[pr * p.ldc + c] = v;
                if (p.Cb) p.Cb[pr * p.ldc + c] = (bf16_t)v;
            }
        }
    }
}

template <typename T>
static void launch_t(const GemmP& p, hipStream_t st) {
    const bool small_n = (p.N % 128) != 0;
    const bool small_m = p.M <= 512;
    if (small_n || small_m) {
        // more, smaller tiles: fills the chip for the batch-1 policy pass and covers N % 128 != 0
        const int grid = ((p.M + 63) / 64) * (p.N / 64);
        hipLaunchKernelGGL((gemm_kernel<T, 64, 64>), dim3(grid), dim3(256), 0, st, p);
    } else {
        const int grid = ((p.M + 127) / 128) * (p.N / 128);
        hipLaunchKernelGGL((gemm_kernel<T, 128, 128>), dim3(grid), dim3(256), 0, st, p);
    }
}

void launch_gemm(const GemmP& p, int dtype, hipStream_t st) {
    if (p.M <= 0) return;
    if (dtype == DT_BF16)
        launch_t<bf16_t>(p, st);
    else
        launch_t<float>(p, st);
}

}  // namespace m3pc
