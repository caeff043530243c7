// The cross-candidate tail of a plan step (learner.py:318-325) and the top-k used by the fp32 re-score.
// Everything here is one workgroup: N <= 16384 scores live in registers / LDS, reductions are 64-lane
// shuffles + one LDS hop.
#include <atomic>
#include <stdlib.h>

#include "kernels.h"

namespace m3pc {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {  // larger value, ties -> lower index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax a, float* sv, int* si) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax b{__shfl_xor(a.v, o), __shfl_xor(a.i, o)};
        a = better(a, b);
    }
    __syncthreads();
    if (lane == 0) {
        sv[wid] = a.v;
        si[wid] = a.i;
    }
    __syncthreads();
    ArgMax r{sv[0], si[0]};
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = better(r, ArgMax{sv[w], si[w]});
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sv) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wsum(v);
    __syncthreads();
    if (lane == 0) sv[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sv[w];
    return t;
}

// p = exp(T (E - max E)) / sum;  eval = sum p a0 / sum p;  argmax E;  sample = argmax p / expo
__device__ __forceinline__ void select_body(const SelectP& p, float* sv, int* si) {
    const int tid = threadIdx.x;
    ArgMax am{-INFINITY, 0x7fffffff};
    for (int i = tid; i < p.n; i += 1024) am = better(am, ArgMax{p.er[i], i});
    am = block_argmax(am, sv, si);
    if (tid == 0 && p.argmax) *p.argmax = am.i;
    const float mx = am.v;
    float s = 0.f;
    for (int i = tid; i < p.n; i += 1024) s += expf(__fmul_rn(__fsub_rn(p.er[i], mx), p.temperature));
    const float tot = block_sum(s, sv);
    float ps = 0.f;
    ArgMax sm{-INFINITY, 0x7fffffff};
    for (int i = tid; i < p.n; i += 1024) {
        const float pi = expf(__fmul_rn(__fsub_rn(p.er[i], mx), p.temperature)) / tot;
        if (p.p) p.p[i] = pi;
        ps += pi;
        if (p.expo) sm = better(sm, ArgMax{pi / p.expo[i], i});
    }
    const float psum = block_sum(ps, sv);
    if (p.expo) {
        sm = block_argmax(sm, sv, si);
        if (tid == 0 && p.sample_idx) *p.sample_idx = sm.i;
        if (p.sample_action && tid < p.A) p.sample_action[tid] = p.a0[(long long)sm.i * p.a0_stride + tid];
    }
    if (p.eval_action) {
        for (int a = 0; a < p.A; ++a) {
            float acc = 0.f;
            for (int i = tid; i < p.n; i += 1024) {
                const float pi = expf(__fmul_rn(__fsub_rn(p.er[i], mx), p.temperature)) / tot;
                acc = fmaf(p.a0[(long long)i * p.a0_stride + a], pi, acc);
            }
            const float t = block_sum(acc, sv);
            if (tid == 0) p.eval_action[a] = t / psum;
        }
    }
}
__global__ __launch_bounds__(1024) void select_kernel(SelectP p) {
    __shared__ float sv[16];
    __shared__ int si[16];
    select_body(p, sv, si);
}
void launch_select(const SelectP& p, hipStream_t st) {
    if (p.n <= 0) return;
    hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, st, p);
}

// ---------------------------------------------------------------------------------------------- top-k
// Bitonic sort of 64-bit keys {orderable(value), ~index} in LDS (descending), first k indices out.
__device__ __forceinline__ unsigned int orderable(float f) {
    unsigned int u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;  // -0.0 orders as +0.0 (a tie, decided by the index), as in a float compare
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// The multinomial draw of learner.py:324-325 as a race: torch.multinomial(p, 1) = argmax_j p_j / q_j, q ~ Exp(1), and with
// p = softmax(tau (E - max E)) that is argmax_j (tau E_j - log q_j).  race_key is that key for one candidate; the certified
// re-score compares keys built from bf16 scores with keys built from fp32 re-scores (m3pc_topk_race_window / _merge_race).
__device__ __forceinline__ float race_key(float tau, float e, float q) { return __fsub_rn(__fmul_rn(tau, e), logf(q)); }

// What a top-k ranks and where the winners go: value[e] = v[e], or race_key(tau, v[e], expo[e]) when expo is given; the
// element of rank r (descending; ties to the lower index) is written to out[r * out_stride].
struct TopSrc {
    const float* v;
    const float* expo;
    float tau;
    int k;
    int* out;
    int out_stride;
    float* scores_out;  // optional: v[winner] beside its index, same stride (rank-by-counting kernels only)
};
__device__ __forceinline__ float src_value(const TopSrc& s, int e) { return s.expo ? race_key(s.tau, s.v[e], s.expo[e]) : s.v[e]; }

__global__ __launch_bounds__(1024) void topk_kernel(const float* v, int n, int npow2, int k, int* idx_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int tid = threadIdx.x;
    for (int i = tid; i < npow2; i += 1024)
        keys[i] = i < n ? ((unsigned long long)orderable(v[i]) << 32) | (unsigned int)(~i) : 0ull;
    __syncthreads();
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (npow2 >> 1); t += 1024) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 1024) idx_out[i] = (int)(~(unsigned int)(keys[i] & 0xffffffffull));
}
// k <= 64 of n <= 16384 by selection instead of a full sort: every wave extracts the k best of its own 1/16 of the
// values with k rounds of (register-local arg-max, 64-lane butterfly) -- no barrier --, then wave 0 does the same over
// the 16 k survivors.  Same order as the bitonic kernel (descending value, ties to the lower index); ~3x faster
// at n = 1024 and independent of n up to 16384 (the replicated top-k of an 8-GPU run sorts 8192 scores).
__global__ __launch_bounds__(1024) void topk_select_kernel(TopSrc src, int n) {
    __shared__ float cv[16 * 64];
    __shared__ int ci[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int k = src.k;
    float val[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int e = s * 1024 + tid;
        val[s] = e < n ? src_value(src, e) : -INFINITY;
    }
    for (int r = 0; r < k; ++r) {
        ArgMax a{-INFINITY, 0x7fffffff};
#pragma unroll
        for (int s = 0; s < 16; ++s) a = better(a, ArgMax{val[s], s * 1024 + tid});
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a = better(a, ArgMax{__shfl_xor(a.v, o), __shfl_xor(a.i, o)});
        if (lane == 0) {
            cv[wid * k + r] = a.v;
            ci[wid * k + r] = a.i;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s * 1024 + tid == a.i) val[s] = -INFINITY;
    }
    __syncthreads();
    if (wid != 0) return;
    float mv[16];
    int mi[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = j * 64 + lane;
        mv[j] = c < 16 * k ? cv[c] : -INFINITY;
        mi[j] = c < 16 * k ? ci[c] : 0x7fffffff;
    }
    for (int r = 0; r < k; ++r) {
        ArgMax a{-INFINITY, 0x7fffffff};
#pragma unroll
        for (int j = 0; j < 16; ++j) a = better(a, ArgMax{mv[j], mi[j]});
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a = better(a, ArgMax{__shfl_xor(a.v, o), __shfl_xor(a.i, o)});
        if (lane == 0) src.out[r * src.out_stride] = a.i;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (mi[j] == a.i) mv[j] = -INFINITY;
    }
}

// n <= 2048: rank by counting.  Every element's 64-bit key {orderable(value), ~index} (the bitonic kernel's, so the order
// is the same: descending value, ties to the lower index) is compared with all n keys, read as LDS broadcasts; keys are
// distinct, so rank = #(greater keys) is a permutation and the elements of rank < k write themselves out.  No shuffle
// chains, no rounds: ~5 us at n = 1024 where the k-round selection below takes 33 (16 rounds x 12 dependent shuffles,
// twice).
// Many-block form of the same ranking (n <= 2048): 256 threads per block, thread t of block b ranks element 64 b + (t & 63)
// against a quarter of the keys (t >> 6), the four partial ranks meet in LDS.  16 blocks at n = 1024 instead of one:
// the n^2 / 2 key comparisons spread over 16 CUs (20 us -> a few).
__global__ __launch_bounds__(256) void topk_rank_blocks_kernel(const float* v, int n, int k, int* idx_out) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[2048];
    __shared__ int part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    for (int e = tid; e < 2048; e += 256) keys[e] = e < n ? ((unsigned long long)orderable(v[e]) << 32) | (unsigned int)(~e) : 0ull;
    __syncthreads();
    const int e = blockIdx.x * 64 + lane;
    const unsigned long long mine = e < n ? keys[e] : 0ull;
    const int n4 = (((n + 3) / 4) + 1) & ~1;  // keys per quarter, even (keys past n are 0: never greater than a real key)
    int rank = 0;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int j0 = q * n4, j1 = j0 + n4 < 2048 ? j0 + n4 : 2048;
#pragma unroll 8
    for (int j = j0; j < j1; j += 2) {
        const u64x2 kj = *(const u64x2*)&keys[j];
        rank += (kj.x > mine) + (kj.y > mine);
    }
    part[q][lane] = rank;
    __syncthreads();
    if (q == 0 && e < n) {
        rank = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        if (rank < k) idx_out[rank] = e;
    }
}

// Two rankings in one launch (n <= 2048): blockIdx.y picks the source -- the scores themselves / the race keys.
__global__ __launch_bounds__(256) void topk_rank_blocks2_kernel(TopSrc s0, TopSrc s1, int n) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[2048];
    __shared__ int part[4][64];
    const TopSrc s = blockIdx.y ? s1 : s0;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    for (int e = tid; e < 2048; e += 256) keys[e] = e < n ? ((unsigned long long)orderable(src_value(s, e)) << 32) | (unsigned int)(~e) : 0ull;
    __syncthreads();
    const int e = blockIdx.x * 64 + lane;
    const unsigned long long mine = e < n ? keys[e] : 0ull;
    const int n4 = (((n + 3) / 4) + 1) & ~1;
    int rank = 0;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int j0 = q * n4, j1 = j0 + n4 < 2048 ? j0 + n4 : 2048;
#pragma unroll 8
    for (int j = j0; j < j1; j += 2) {
        const u64x2 kj = *(const u64x2*)&keys[j];
        rank += (kj.x > mine) + (kj.y > mine);
    }
    part[q][lane] = rank;
    __syncthreads();
    if (q == 0 && e < n) {
        rank = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        if (rank < s.k) {
            s.out[rank * s.out_stride] = e;
            if (s.scores_out) s.scores_out[rank * s.out_stride] = s.v[e];
        }
    }
}

// The same two rankings for 2048 < n <= 16384 (config 3's 4096 candidates, the gathered scores of a candidate-sharded config 4):
// the keys of ALL n elements in dynamic LDS (8 n bytes: 32 KB at 4096, 128 KB at 16384), 64 elements per block ranked by 256
// threads, a quarter of the keys each.  One launch of ~10-20 us where the selection kernel needs k rounds of butterflies in a
// single workgroup (~60 us for the 32 racers) and the 129 best by score a whole bitonic sort.
__global__ __launch_bounds__(256) void topk_rank_blocks2_dyn_kernel(TopSrc s0, TopSrc s1, int n, int n4) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long dkeys[];  // 4 * n4 entries
    __shared__ int part[4][64];
    const TopSrc s = blockIdx.y ? s1 : s0;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    for (int e = tid; e < 4 * n4; e += 256) dkeys[e] = e < n ? ((unsigned long long)orderable(src_value(s, e)) << 32) | (unsigned int)(~e) : 0ull;
    __syncthreads();
    const int e = blockIdx.x * 64 + lane;
    const unsigned long long mine = e < n ? dkeys[e] : 0ull;
    int rank = 0;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int j0 = q * n4, j1 = j0 + n4;
#pragma unroll 8
    for (int j = j0; j < j1; j += 2) {
        const u64x2 kj = *(const u64x2*)&dkeys[j];
        rank += (kj.x > mine) + (kj.y > mine);
    }
    part[q][lane] = rank;
    __syncthreads();
    if (q == 0 && e < n) {
        rank = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        if (rank < s.k) {
            s.out[rank * s.out_stride] = e;
            if (s.scores_out) s.scores_out[rank * s.out_stride] = s.v[e];
        }
    }
}

__global__ __launch_bounds__(1024) void topk_rank_kernel(const float* v, int n, int k, int* idx_out) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[2048];
    const int tid = threadIdx.x;
    unsigned long long mine[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int e = s * 1024 + tid;
        mine[s] = e < n ? ((unsigned long long)orderable(v[e]) << 32) | (unsigned int)(~e) : 0ull;
        keys[e] = mine[s];
    }
    __syncthreads();
    const int n2 = (n + 1) & ~1;  // (keys past n are 0: never greater than a real key)
    int rank0 = 0, rank1 = 0;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    if (n <= 1024) {
#pragma unroll 8
        for (int j = 0; j < n2; j += 2) {
            const u64x2 kj = *(const u64x2*)&keys[j];
            rank0 += (kj.x > mine[0]) + (kj.y > mine[0]);
        }
    } else {
#pragma unroll 4
        for (int j = 0; j < n2; j += 2) {
            const u64x2 kj = *(const u64x2*)&keys[j];
            rank0 += (kj.x > mine[0]) + (kj.y > mine[0]);
            rank1 += (kj.x > mine[1]) + (kj.y > mine[1]);
        }
    }
    if (tid < n && rank0 < k) idx_out[rank0] = tid;
    if (1024 + tid < n && rank1 < k) idx_out[rank1] = 1024 + tid;
}

void launch_topk(const float* v, int n, int k, int* idx_out, hipStream_t st) {
    if (n <= 0 || k <= 0) return;
    if (n <= 2048 && k <= n) {
        static const bool one_block = M3PC_ENV("M3PC_TOPK_ONE_BLOCK") != nullptr;  // A/B switch
        if (one_block) hipLaunchKernelGGL(topk_rank_kernel, dim3(1), dim3(1024), 0, st, v, n, k, idx_out);
        else hipLaunchKernelGGL(topk_rank_blocks_kernel, dim3((n + 63) / 64), dim3(256), 0, st, v, n, k, idx_out);
        return;
    }
    if (k <= 64 && n <= 16384) {
        hipLaunchKernelGGL(topk_select_kernel, dim3(1), dim3(1024), 0, st, TopSrc{v, nullptr, 0.f, k, idx_out, 1, nullptr}, n);
        return;
    }
    int np = 2;
    while (np < n) np <<= 1;
    static bool attr[64] = {};  // (a per-device opt-in)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr[dev]) {
        (void)hipFuncSetAttribute((const void*)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
        attr[dev] = true;
    }
    hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), (size_t)np * 8, st, v, n, np, k, idx_out);
}

// The two candidate lists of the certified re-score in one buffer (m3pc_topk_race_window): list[rmax + i] = the i-th best
// entry of v (i <= kk - 1), list[rmax - 1 - i] = the i-th best entry by race key (i < rr): the r best racers and the n best
// scorers are the contiguous slice [rmax - r, rmax + n).
bool launch_topk_race(const float* v, const float* expo, float tau, int n, int kk, int rr, int rmax, int* list, float* list_scores,
                      hipStream_t st) {
    if (n <= 0 || kk <= 0) return true;
    const bool one = n <= 16384;  // rank by counting: every element knows its value and its rank -- the scores go out with the ids
    const TopSrc s0{v, nullptr, 0.f, kk, list + rmax, 1, one && list_scores ? list_scores + rmax : nullptr};
    const TopSrc s1{v, expo, tau, rr, list + rmax - 1, -1, one && list_scores ? list_scores + rmax - 1 : nullptr};
    if (n <= 2048) {
        hipLaunchKernelGGL(topk_rank_blocks2_kernel, dim3((n + 63) / 64, rr > 0 ? 2 : 1), dim3(256), 0, st, s0, s1, n);
        return true;  // (list_scores written)
    }
    if (one) {  // n <= 16384: the keys in dynamic LDS
        const int n4 = (((n + 3) / 4) + 1) & ~1;
        const size_t lds = (size_t)4 * n4 * sizeof(unsigned long long);
        // > 64 KiB of dynamic LDS is a per-device opt-in, asked for ONCE per device and only when the launch needs it (n > 8192);
        // a refusal is remembered and its error cleared, so that the fallback below is not reported as a failed launch by the
        // caller's hipGetLastError (the same pattern as lds_opt_in in attn_bf16.hip)
        static std::atomic<int> state[64] = {};  // 0: not asked, 1: granted, 2: refused
        int dev = 0;
        const bool dev_ok = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        bool fits = lds <= 64 * 1024;
        if (!fits && dev_ok) {
            int stt = state[dev].load(std::memory_order_acquire);
            if (stt == 0) {
                stt = hipFuncSetAttribute((const void*)topk_rank_blocks2_dyn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024) == hipSuccess ? 1 : 2;
                if (stt == 2) (void)hipGetLastError();
                state[dev].store(stt, std::memory_order_release);
            }
            fits = stt == 1;
        }
        if (fits) {
            hipLaunchKernelGGL(topk_rank_blocks2_dyn_kernel, dim3((n + 63) / 64, rr > 0 ? 2 : 1), dim3(256), lds, st, s0, s1, n, n4);
            return true;
        }
    }
    // (the opt-in was refused: the selection kernels, no scores)
    const TopSrc t1{v, expo, tau, rr, list + rmax - 1, -1, nullptr};
    launch_topk(v, n, kk, list + rmax, st);
    if (rr > 0) hipLaunchKernelGGL(topk_select_kernel, dim3(1), dim3(1024), 0, st, t1, n);  // (rr <= 64, n <= 16384)
    return false;     // (the caller still has to gather list_scores: launch_window_stats)
}

// idx: the kk best entries of v, best first.  n = clamp(#{i < kk: v[idx[i]] >= v[idx[0]] - window}, kmin, kmax) and the
// distance from the best entry to the best one NOT among those n (infinity when there is none).
// Window statistics of the bound-driven re-score: idx = the kk best entries of v (best first).  The raw count of entries
// within `window` of the maximum runs over the WHOLE vector (n_total), so a caller sees when the window holds more than the
// kmax entries it may list (stats[3] > kmax: the listed prefix no longer covers the window).
__global__ __launch_bounds__(256) void window_stats_kernel(const float* v, int n_total, const int* idx, int kk, int kmin, int kmax,
                                                           float window, float* stats, float* host_stats, float seq,
                                                           float* top_scores, int rr) {
    __shared__ int part[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float mx = v[idx[0]];
    int cnt = 0;
    for (int i = tid; i < n_total; i += 256) cnt += v[i] >= mx - window ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) part[wid] = cnt;
    if (top_scores)
        for (int i = tid - rr; i < kk; i += 256) top_scores[i] = v[idx[i]];  // (entries -rr .. -1: the race list in front of idx)
    __syncthreads();
    cnt = part[0] + part[1] + part[2] + part[3];
    int n = cnt < kmin ? kmin : cnt;
    if (n > kmax) n = kmax;
    if (n > kk) n = kk;
    if (tid == 0) {
        const float margin = n < kk ? mx - v[idx[n]] : INFINITY;
        stats[0] = (float)n;
        stats[1] = margin;
        stats[2] = mx;
        stats[3] = (float)cnt;
        if (host_stats) {  // host-mapped (pinned) copy the caller spins on: payload first, then the sequence number
            host_stats[0] = (float)n;
            host_stats[1] = margin;
            host_stats[2] = mx;
            host_stats[3] = (float)cnt;
            __threadfence_system();
            __hip_atomic_store(host_stats + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
void launch_window_stats(const float* v, int n_total, const int* idx, int kk, int kmin, int kmax, float window, float* stats,
                         float* host_stats, float seq, float* top_scores, hipStream_t st, int rr) {
    hipLaunchKernelGGL(window_stats_kernel, dim3(1), dim3(256), 0, st, v, n_total, idx, kk, kmin, kmax, window, stats, host_stats, seq,
                       top_scores, rr);
}

// Merge of a bf16 score vector with the fp32 re-scores of the listed candidates (planner: certified re-score).  The list holds
// r race entries (the r best by race key, any order) followed by n score entries (the n best by bf16 score, best first); m = r + n.
//   d_i = b[list[i]] - f[i] over the m listed entries;  c = lower median(d): the common shift of the bf16 scores;
//   out[j] = b[j] - c for every j, then out[list[i]] = f[i] (score entries first, race entries behind them: deterministic when a
//   candidate sits in both parts).
// Arg-max certificate: with |(b_j - f_j) - c| <= delta for every candidate, an un-listed j can only beat the best listed fp32 score
// f* if b_j - c > f* - delta.  need = #{j : b[j] > f* + c - delta} over the WHOLE vector (the score entries are the n largest b, so
// this set is a prefix of the descending order): need <= n certifies that the arg-max of `out` is the fp32 arg-max; otherwise the
// entries n .. need-1 of the order still have to be re-scored.
// Race certificate (expo given): K* = max over the listed entries of race_key(tau, f_i, q_i) is a key the multinomial's winner at
// least reaches; an un-listed j can only win if tau (b_j - c + delta) - log q_j >= K*, i.e. race_key(tau, b_j, q_j) >= K* +
// tau (c - delta).  need_race = the number of such j over the WHOLE vector -- a prefix of the descending race-key order -- so
// need_race <= r certifies that arg-max_j p_j / q_j over `out` is the fp32 draw; otherwise the race entries r .. need_race-1
// still have to be re-scored.
//   stats = {c, max_i |d_i - c|, need, (f* + c - delta) - (largest un-listed b) [inf when everything is listed], -, need_race,
//            K*, K* + tau (c - delta)}   (slot 4 of the host copy carries the sequence number)
__device__ __forceinline__ void rescore_merge_body(const float* b, int n_total, const int* list, int r, int n, const float* list_scores, const float* f, float delta, const float* expo, float tau, float* out, float* stats, float* host_stats, float seq, int nstats) {
    __shared__ float d[1024];
    __shared__ int ids[1024];
    __shared__ float c_sh;
    __shared__ float sv[16], sf[16], sb[16], sk[16];
    __shared__ int sc[16], sr[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = r + n;
    float kf = -INFINITY;
    if (tid < m) {
        d[tid] = list_scores[tid] - f[tid];
        ids[tid] = list[tid];
        if (expo) kf = race_key(tau, f[tid], expo[ids[tid]]);
    }
    __syncthreads();
    if (tid < m) {
        const float me = d[tid];
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const float o = d[j];
            rank += (o < me || (o == me && j < tid)) ? 1 : 0;
        }
        if (rank == (m - 1) / 2) c_sh = me;
    }
    __syncthreads();
    const float c = c_sh;
    float dev = tid < m ? fabsf(d[tid] - c) : 0.f;
    float fb = tid < m ? f[tid] : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dev = fmaxf(dev, __shfl_xor(dev, o));
        fb = fmaxf(fb, __shfl_xor(fb, o));
        kf = fmaxf(kf, __shfl_xor(kf, o));
    }
    if (lane == 0) {
        sv[wid] = dev;
        sf[wid] = fb;
        sk[wid] = kf;
    }
    __syncthreads();
    float fbest = sf[0], devmax = sv[0], kbest = sk[0];
    for (int w = 1; w < 16; ++w) {
        fbest = fmaxf(fbest, sf[w]);
        devmax = fmaxf(devmax, sv[w]);
        kbest = fmaxf(kbest, sk[w]);
    }
    const float thr = fbest + c - delta;            // on the bf16 scale
    const float thr_r = kbest + tau * (c - delta);  // on the bf16 race-key scale
    int cnt = 0, cnt_r = 0;
    float bout = -INFINITY;  // largest b among the un-listed: everything below the n-th largest b (the score entries are a prefix)
    const float b_last = list_scores[m - 1];
    for (int j = tid; j < n_total; j += 1024) {
        const float v = b[j];
        out[j] = v - c;
        cnt += v > thr ? 1 : 0;
        if (v < b_last) bout = fmaxf(bout, v);
        if (expo) cnt_r += race_key(tau, v, expo[j]) >= thr_r ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o);
        cnt_r += __shfl_xor(cnt_r, o);
        bout = fmaxf(bout, __shfl_xor(bout, o));
    }
    if (lane == 0) {
        sc[wid] = cnt;
        sr[wid] = cnt_r;
        sb[wid] = bout;
    }
    __syncthreads();
    if (tid >= r && tid < m) out[ids[tid]] = f[tid];
    __syncthreads();
    if (tid < r) out[ids[tid]] = f[tid];
    if (tid == 0) {
        int need = 0, need_r = 0;
        float bo = -INFINITY;
        for (int w = 0; w < 16; ++w) {
            need += sc[w];
            need_r += sr[w];
            bo = fmaxf(bo, sb[w]);
        }
        // (ties with the n-th listed score count as listed above: a tie at the list's edge keeps `need` honest through cnt)
        const float margin = n >= n_total ? INFINITY : thr - bo;
        const float st8[8] = {c, devmax, (float)need, margin, 0.f, (float)need_r, kbest, thr_r};
        for (int i = 0; i < nstats; ++i) stats[i] = st8[i];
        if (host_stats) {
            // all 8 host slots on every merge: a four-statistics merge (no race list) leaves zeros at 5..7 instead of the
            // previous race merge's need_race / K* / threshold on the same slot
            for (int i = 0; i < 8; ++i)
                if (i != 4) host_stats[i] = i < nstats ? st8[i] : 0.f;
            __threadfence_system();
            __hip_atomic_store(host_stats + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(1024) void rescore_merge_kernel(const float* b, int n_total, const int* list, int r, int n, const float* list_scores, const float* f, float delta, const float* expo, float tau, float* out, float* stats, float* host_stats, float seq, int nstats) {
    rescore_merge_body(b, n_total, list, r, n, list_scores, f, delta, expo, tau, out, stats, host_stats, seq, nstats);
}
// merge + select in one launch (both are one workgroup over the whole vector): the select reads the merged vector the
// workgroup itself has just written (workgroup-scope visibility behind __syncthreads)
__global__ __launch_bounds__(1024) void merge_select_kernel(const float* b, int n_total, const int* list, int r, int n, const float* list_scores, const float* f, float delta, const float* expo, float tau, float* out, float* stats, float* host_stats, float seq, int nstats, SelectP sp) {
    __shared__ float sv[16];
    __shared__ int si[16];
    rescore_merge_body(b, n_total, list, r, n, list_scores, f, delta, expo, tau, out, stats, host_stats, seq, nstats);
    __syncthreads();
    select_body(sp, sv, si);
}
void launch_rescore_merge(const float* b, int n_total, const int* list, int r, int n, const float* list_scores, const float* f,
                          float delta, const float* expo, float tau, float* out, float* stats, float* host_stats, float seq,
                          hipStream_t st) {
    hipLaunchKernelGGL(rescore_merge_kernel, dim3(1), dim3(1024), 0, st, b, n_total, list, r, n, list_scores, f, delta, expo, tau, out,
                       stats, host_stats, seq, expo ? 8 : 4);
}
void launch_merge_select(const float* b, int n_total, const int* list, int r, int n, const float* list_scores, const float* f,
                         float delta, const float* expo, float tau, float* out, float* stats, float* host_stats, float seq,
                         const SelectP& sp, hipStream_t st) {
    hipLaunchKernelGGL(merge_select_kernel, dim3(1), dim3(1024), 0, st, b, n_total, list, r, n, list_scores, f, delta, expo, tau, out,
                       stats, host_stats, seq, expo ? 8 : 4, sp);
}

__global__ void scatter_kernel(const float* src, const int* index, int n, float* dst, int* index_copy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dst[index[i]] = src[i];
        if (index_copy) index_copy[i] = index[i];
    }
}
void launch_scatter(const float* src, const int* index, int n, float* dst, int* index_copy, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, index, n, dst, index_copy);
}

}  // namespace m3pc
