// Which bf16 MFMA shape should the fused layer tail (m3pc_amd/csrc/block_fused.hip) run on?  VERDICT r4 item 3a.
// MI355X_MICROARCH.md (DVFS give-back, item 7): where the chip holds its clock down under load, the clock it holds depends on the
// MFMA shape -- a bare v_mfma_f32_16x16x32_bf16 loop delivered 1.12-1.15 x the FLOP/s of the 32x32x16 loop at equal cycles per FLOP.
// The fused tail is not a bare loop: ONE wave per SIMD issues, per 1-KiB weight fragment (= 32 matrix cycles either way), one
// ds_read_b128, its share of the gelu's VALU work and of the LDS-DMA pieces beside the MFMA(s), and an MFMA holds the SIMD's vector
// issue for 8 of its cycles -- 8 of 32 for one 32x32x16, 16 of 32 for the two 16x16x32 that consume the same fragment.
// This probe runs both shapes as that kernel would: same per-wave output tile (32 tokens x 512 features in 256 accumulator
// registers... scaled to 8 feature tiles = 128 registers here), weights = A operand re-read from LDS by ds_read_b128 (one fragment
// per 32 matrix cycles, four fragments ahead), activations = B operand in registers, random data, one wave per SIMD, every CU busy;
// FILL independent v_fma_f32 per fragment stand for the gelu (the kernel's FFN phases carry ~6 VALU + 2 transcendentals per MFMA
// in half of their phases).  Prints wall time, TFLOP/s, in-kernel cycles per fragment and the clock the chip held.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe && tools/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NFRAG 64  // fragments of the LDS weight image one pass walks (64 KiB)

__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// NW: waves per workgroup (4: one per SIMD, as the fused tail runs; 8: two per SIMD -- round 6, VERDICT r5 item 1: what would a second
// wave per SIMD buy an LDS-fed MFMA loop of this filler density on this chip, in cycles AND in wall time?  Per-wave state is the same
// 128 accumulator registers either way, so NW = 8 does twice the work per workgroup).  BAR: one s_barrier per 32 fragments per wave
// (the ring-stage sync of the fused tail).
template <int SHAPE, int FILL, int NW = 4, int BAR = 0>
__global__ __launch_bounds__(64 * NW) void probe(float* out, long long* stamps, int passes, const unsigned short* seed_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // random weight image (bf16 bit patterns of values in [-1, 1)), the same for every workgroup
    for (int i = tid; i < NFRAG * 512; i += 64 * NW) ((unsigned short*)lds)[i] = seed_bits[i];
    __syncthreads();
    bf16x8 b[8];  // the wave's activations: 8 k-steps of its 32 token rows (B operand; for 16x16x32 two 16-token groups per register set)
    for (int j = 0; j < 8; ++j)
        for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(((lane * 8 + e + 13 * j) % 97) * (1.0f / 97.0f) - 0.5f);
    f32x16 acc[8];
    f32x4 acc4[16];
    for (int j = 0; j < 8; ++j) acc[j] = (f32x16){0};
    for (int j = 0; j < 16; ++j) acc4[j] = (f32x4){0};
    float fz[8];
    for (int j = 0; j < 8; ++j) fz[j] = 0.001f * (lane + j);
    const unsigned base = lane * 16;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    bf16x8 cur[4], nxt[4];
    for (int q = 0; q < 4; ++q) cur[q] = lds_frag(base + q * 1024);
    for (int p = 0; p < passes; ++p) {
#pragma unroll
        for (int g = 0; g < NFRAG / 4; ++g) {
            // four fragments ahead: the reads of the next group are issued before this group's MFMAs
            const unsigned nb = base + (((g + 1) & (NFRAG / 4 - 1)) * 4) * 1024;
            if (BAR && (g & 7) == 0) asm volatile("s_barrier" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
#pragma unroll
            for (int q = 0; q < 4; ++q) nxt[q] = lds_frag(nb + q * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = (g * 4 + q) & 7;
                if (SHAPE == 32) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[q], b[t], acc[t], 0, 0, 0);
                } else {
                    // the same 1-KiB fragment (16 features x 32 k) against the two 16-token groups: two accumulators of 4 registers
                    acc4[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[q], b[t], acc4[2 * t], 0, 0, 0);
                    acc4[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[q], b[(t + 1) & 7], acc4[2 * t + 1], 0, 0, 0);
                }
#pragma unroll
                for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fz[(f + q) & 7]) : "v"(fz[(f + q + 3) & 7]), "v"(0.5f));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int j = 0; j < 8; ++j) {
        for (int i = 0; i < 16; ++i) s += acc[j][i];
        s += fz[j];
    }
    for (int j = 0; j < 16; ++j) s += acc4[j][0] + acc4[j][1] + acc4[j][2] + acc4[j][3];
    for (int q = 0; q < 4; ++q) s += (float)cur[q][0];
    out[blockIdx.x * 64 * NW + tid] = s;
    if (lane == 0) {
        stamps[(blockIdx.x * NW + (tid >> 6)) * 2] = t1 - t0;
        stamps[(blockIdx.x * NW + (tid >> 6)) * 2 + 1] = r1 - r0;
    }
}

template <int SHAPE, int FILL, int NW = 4, int BAR = 0>
static void run(float* out, long long* stamps, const unsigned short* seed, int passes) {
    const int blocks = 256;
    passes = passes * 4 / NW;  // (the same work per workgroup)
    hipFuncSetAttribute((const void*)probe<SHAPE, FILL, NW, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, NFRAG * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 40; ++w) probe<SHAPE, FILL, NW, BAR><<<blocks, 64 * NW, NFRAG * 1024>>>(out, stamps, passes, seed);  // warm: the clock settles under load
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<SHAPE, FILL, NW, BAR><<<blocks, 64 * NW, NFRAG * 1024>>>(out, stamps, passes, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> st(blocks * NW * 2);
    hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int i = 0; i < blocks * NW; ++i) {
        cyc.push_back((double)st[2 * i]);
        clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);  // s_memrealtime ticks at 100 MHz -> GHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double frags = (double)passes * NFRAG;
    const double flops = (double)blocks * NW * frags * 2.0 * 32 * 32 * 16;
    // cycles per fragment PER SIMD: with two waves per SIMD a wave's own fragment takes twice the SIMD's
    printf("shape %s fill %d waves/SIMD %d barrier %d: wall %.3f ms  %.1f TFLOP/s  cycles/fragment/SIMD %.2f (median wave)  in-kernel clock %.3f GHz\n",
           SHAPE == 32 ? "32x32x16" : "16x16x32", FILL, NW / 4, BAR, ms, flops / ms / 1e9, cyc[cyc.size() / 2] / frags / (NW / 4),
           clk[clk.size() / 2]);
}

int main() {
    float* out;
    long long* stamps;
    unsigned short* seed;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&stamps, 256 * 8 * 2 * 8);
    hipMalloc(&seed, NFRAG * 1024);
    std::vector<unsigned short> h(NFRAG * 512);
    srand(1);
    for (auto& v : h) {
        const float f = (rand() / (float)RAND_MAX) * 2.f - 1.f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    hipMemcpy(seed, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int passes = 4000;  // 256000 fragments per wave: ~5 ms per launch
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 0>(out, stamps, seed, passes);
        run<16, 0>(out, stamps, seed, passes);
        run<32, 3>(out, stamps, seed, passes);
        run<16, 3>(out, stamps, seed, passes);
        run<32, 6>(out, stamps, seed, passes);
        run<16, 6>(out, stamps, seed, passes);
        // round 6: a second wave per SIMD (and the per-stage barrier of a shared weight ring)
        run<32, 0, 8>(out, stamps, seed, passes);
        run<32, 3, 8>(out, stamps, seed, passes);
        run<32, 6, 8>(out, stamps, seed, passes);
        run<32, 3, 4, 1>(out, stamps, seed, passes);
        run<32, 3, 8, 1>(out, stamps, seed, passes);
        run<32, 6, 4, 1>(out, stamps, seed, passes);
        run<32, 6, 8, 1>(out, stamps, seed, passes);
        run<16, 3, 8>(out, stamps, seed, passes);
        run<16, 6, 8>(out, stamps, seed, passes);
    }
    return 0;
}
