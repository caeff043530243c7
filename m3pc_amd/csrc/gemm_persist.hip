// Persistent bf16 MFMA GEMM for the short-K (K = 512 / 2048), many-row GEMMs of the candidate pass.
//
// 256x256 tiles, 8 waves (2 x 4, wave tile 128 x 64), 64-deep K stages in a 2-slot LDS ring filled by LDS-DMA,
// hand-scheduled stage (gemm_stage_asm.h) -- the same tile program as gemm_glds.hip -- but ONE workgroup per CU
// walks a list of tiles and the K-stage stream runs ACROSS tile boundaries:
//     compute(stage s) ; barrier ; issue(stage s+2) ; [tile finished: epilogue]
// so (a) the first stages of the next tile are already in LDS / in flight while the current tile's epilogue
// stores, (b) there is no per-tile block launch, LDS allocation and cold prologue.  With K = 512 a tile is only 8
// stages long, and those fixed costs were ~27 % of a tile's time in the one-tile-per-block kernel.
// Tile order: virtual block id v = block + i * grid; v % 8 labels the XCD (round-robin dispatch), every XCD
// walks its own contiguous range of tiles, so the column tiles of one A row-panel meet in one L2.
#include "gemm_epilogue.h"
#include "gemm_stage_asm.h"
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

enum { EPI_GELU = 1, EPI_RES = 2, EPI_ROWTAB = 4, EPI_F32OUT = 8 };

__device__ __forceinline__ float gelu_fast4(float x) {
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
__device__ __forceinline__ int map_row4(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_persist_kernel(GemmP p) {
    constexpr int BM = 256, BN = 256, WN = 4, WTM = 128, WTN = 64, TM = 4, TN = 2, NW = 8;
    constexpr int NI = 4;  // 1-KiB LDS-DMA instructions per wave per operand per stage
    constexpr int BUF = (BM + BN) * 128;
    __shared__ __attribute__((aligned(1024))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ntn = p.N / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntm * ntn;
    const int nkt = p.K / 64;
    const long long lda_b = (long long)p.lda * 2, ldw_b = (long long)p.ldw * 2;
    const int G = gridDim.x;

    // tile of this block's i-th turn (XCD-contiguous order, bijective for any ntiles)
    auto tile_of = [&](int i) -> int {
        const int v = blockIdx.x + i * G;
        if (v >= ntiles) return -1;
        const int q = ntiles / 8, r = ntiles % 8, x = v % 8, j = v / 8;
        return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    };
    // per-lane DMA source pointers of a tile (stage 0); LDS chunk position c of row r <- logical chunk c ^ ((r>>1)&7)
    auto make_src = [&](int tile, const char** a_src, const char** w_src) {
        const int row0 = (tile / ntn) * BM, col0 = (tile % ntn) * BN;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = 8 * (wid + NW * i) + (lane >> 3);
            const int q = (lane & 7) ^ ((r >> 1) & 7);
            int gr = row0 + r;
            if (gr >= p.M) gr = p.M - 1;
            a_src[i] = (const char*)p.A + (long long)map_row4(p.amap, gr) * lda_b + q * 16;
            w_src[i] = (const char*)p.W + (long long)(col0 + r) * ldw_b + q * 16;
        }
    };
    const int wave_dst = __builtin_amdgcn_readfirstlane(wid) * 1024;
    auto issue = [&](const char* const* a_src, const char* const* w_src, int kt, int buf) {
        char* base = smem + buf * BUF + wave_dst;
        const long long ko = (long long)kt * 128;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + ko), (lptr_t)(base + i * NW * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + ko), (lptr_t)(base + BM * 128 + i * NW * 1024), 16, 0, 0);
    };

    // fragment read offsets (same swizzle as the DMA side)
    const int sw = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = l31 * 128 + (((2 * s + lh) ^ sw) * 16);
    const unsigned ldsb = (unsigned)(size_t)(lptr_t)smem;
    const unsigned fragA = ldsb + wr * WTM * 128, fragW = ldsb + BM * 128 + wc * WTN * 128;

    int turn = 0;
    int tile = tile_of(0);
    if (tile < 0) return;
    const char *a_cur[NI], *w_cur[NI], *a_nxt[NI], *w_nxt[NI];
    make_src(tile, a_cur, w_cur);
    int tile_next = tile_of(1);
    if (tile_next >= 0) make_src(tile_next, a_nxt, w_nxt);

    // stream stages 0 and 1 (nkt >= 2 guaranteed by the launcher)
    issue(a_cur, w_cur, 0, 0);
    issue(a_cur, w_cur, 1, 1);
    __syncthreads();

    f32x16 acc[TM][TN];
    int s = 0;  // stream stage counter (buffer = s & 1)
    for (;;) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < nkt; ++kt, ++s) {
            const unsigned lb = (unsigned)((s & 1) * BUF);
            const unsigned aA0 = fragA + lb + foff[0], aA1 = fragA + lb + foff[1], aA2 = fragA + lb + foff[2],
                           aA3 = fragA + lb + foff[3];
            const unsigned aW0 = fragW + lb + foff[0], aW1 = fragW + lb + foff[1], aW2 = fragW + lb + foff[2],
                           aW3 = fragW + lb + foff[3];
            u32x4 t0, t1, t2, t3, t4, t5, u0, u1, u2, u3, u4, u5;
            asm volatile(M3PC_STAGE_ASM_4x2
                         : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                           "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
                           "=&v"(t4), "=&v"(t5), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5)
                         : "v"(aA0), "v"(aA1), "v"(aA2), "v"(aA3), "v"(aW0), "v"(aW1), "v"(aW2), "v"(aW3)
                         : "memory");
            if (kt + 1 == nkt) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU readers
            __syncthreads();  // every wave is done with buffer s&1, and stage s+1 has landed (vmcnt drained)
            // refill buffer s&1 with stream stage s+2: this tile's kt+2, or the next tile's stage 0 / 1
            if (kt + 2 < nkt)
                issue(a_cur, w_cur, kt + 2, s & 1);
            else if (tile_next >= 0)
                issue(a_nxt, w_nxt, kt + 2 - nkt, s & 1);
        }

        // ---- epilogue of `tile` (the next tile's first two stages are landed / in flight meanwhile)
        const int row0 = (tile / ntn) * BM, col0 = (tile % ntn) * BN;
        {
            const int wu = __builtin_amdgcn_readfirstlane(wid);
            gemm_epilogue<EPI, TM, TN>(p, acc, row0 + (wu / WN) * WTM, col0 + (wu % WN) * WTN, row0, BM, lane);
        }

        // ---- rotate to the next tile
        tile = tile_next;
        if (tile < 0) break;
        ++turn;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            a_cur[i] = a_nxt[i];
            w_cur[i] = w_nxt[i];
        }
        tile_next = tile_of(turn + 1);
        if (tile_next >= 0) make_src(tile_next, a_nxt, w_nxt);
    }
}

template <int EPI>
static void launch_cfg(const GemmP& p, int grid, hipStream_t st) {
    hipLaunchKernelGGL((gemm_persist_kernel<EPI>), dim3(grid), dim3(512), 0, st, p);
}

// returns false when the shape / epilogue is not covered (the caller falls back to the other kernels)
bool launch_gemm_persist(const GemmP& p, hipStream_t st) {
    if (p.K % 64 != 0 || p.K < 128 || p.N % 256 != 0) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 8) || (p.ldw % 8)) return false;
    const long long ntiles = (long long)((p.M + 255) / 256) * (p.N / 256);
    if (ntiles < 256) return false;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        n_cu = prop.multiProcessorCount;
    }
    const int grid = (int)(ntiles < n_cu ? ntiles : n_cu);
    const bool f32out = p.Cf != nullptr;
    const int epi = (p.gelu ? EPI_GELU : 0) | (p.res ? EPI_RES : 0) | (p.rowtab ? EPI_ROWTAB : 0) | (f32out ? EPI_F32OUT : 0);
    switch (epi) {
        case 0: launch_cfg<0>(p, grid, st); return true;
        case EPI_F32OUT: launch_cfg<EPI_F32OUT>(p, grid, st); return true;
        case EPI_GELU: launch_cfg<EPI_GELU>(p, grid, st); return true;
        case EPI_GELU | EPI_F32OUT: launch_cfg<EPI_GELU | EPI_F32OUT>(p, grid, st); return true;
        case EPI_RES | EPI_F32OUT: launch_cfg<EPI_RES | EPI_F32OUT>(p, grid, st); return true;
        case EPI_ROWTAB | EPI_F32OUT: launch_cfg<EPI_ROWTAB | EPI_F32OUT>(p, grid, st); return true;
        default: return false;
    }
}

}  // namespace m3pc
