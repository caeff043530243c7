// What does a grid-wide barrier cost on MI355X?  (the question behind one persistent kernel for the few-row fp32 chains: a chain
// is ~30 dependent launches of 5-10 us each; a barrier replaces a launch boundary)
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
// Every workgroup writes a value before barrier k and reads its neighbour's (another XCD's, by round-robin dispatch) after it:
// release = __threadfence() in front of the arrive, acquire behind the wait (agent scope: L2 write-back / invalidate across XCDs).
// A wait gives up after a bounded number of polls, so the grid always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int sleep) {
    bool ok = true;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(ctr, 1u);
        int polls = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (sleep) __builtin_amdgcn_s_sleep(1);
            if (++polls > (1 << 22)) { ok = false; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void probe(unsigned* ctr, float* slots, int nb, int sleep, int work, long long* out, int* bad) {
    const int G = gridDim.x, b = blockIdx.x;
    long long t0 = wall_clock64();
    float acc = 0.f;
    for (int k = 0; k < nb; ++k) {
        if (work) {  // each thread writes one float of this workgroup's 1-KiB slot
            slots[(size_t)(k & 1) * G * 256 + (size_t)b * 256 + threadIdx.x] = (float)(k * 1000 + b);
        } else if (threadIdx.x == 0) {
            slots[(size_t)(k & 1) * G * 256 + (size_t)b * 256] = (float)(k * 1000 + b);
        }
        if (!grid_barrier(ctr, (unsigned)(G * (k + 1)), sleep)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
        const int nb_ = (b + 1) % G;
        const float v = __builtin_nontemporal_load(&slots[(size_t)(k & 1) * G * 256 + (size_t)nb_ * 256 + (work ? threadIdx.x : 0)]);
        if (v != (float)(k * 1000 + nb_) && threadIdx.x == 0) atomicAdd(bad, 1);
        acc += v;
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[b] = t1 - t0;
        if (acc == -1.f) out[b] = 0;
    }
}

int main() {
    unsigned* ctr;
    float* slots;
    long long* out;
    int* bad;
    hipMalloc(&ctr, 4);
    hipMalloc(&slots, 2 * 1024 * 256 * 4);
    hipMalloc(&out, 1024 * 8);
    hipMalloc(&bad, 4);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
    const int nb = 200;
    for (int work = 0; work < 2; ++work)
        for (int sleep = 0; sleep < 2; ++sleep)
            for (int G : {32, 64, 128, 256, 512}) {
                double best = 1e30;
                int nbad = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(ctr, 0, 4);
                    hipMemset(bad, 0, 4);
                    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, ctr, slots, nb, sleep, work, out, bad);
                    hipDeviceSynchronize();
                    std::vector<long long> h(G);
                    hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost);
                    hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
                    long long mx = 0;
                    for (long long v : h) mx = v > mx ? v : mx;
                    const double us = (double)mx / rate * 1e3 / nb;
                    best = us < best ? us : best;
                }
                printf("workgroups %4d  sleep %d  1KiB-per-wg %d : %.2f us per barrier  (stale or timed-out reads: %d)\n", G, sleep, work, best, nbad);
            }
    return 0;
}
