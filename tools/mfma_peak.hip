// Calibration: sustained v_mfma_f32_32x32x16_bf16 rate of this device with operands in registers
// (no LDS, no memory): the clock-limited ceiling every GEMM number should be read against.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f - i * 0.01f * threadIdx.x); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc = 4; wpc <= 8; wpc += 4) {
        const int iters = 20000, blocks = 256 * (wpc == 4 ? 1 : 1), threads = wpc * 64;
        mfma_loop<<<blocks, threads>>>(out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_loop<<<blocks, threads>>>(out, iters, 0.37f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * wpc * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("waves/CU %d: %.3f ms  %.1f TFLOP/s  (implied clock if 1 MFMA/32cyc/SIMD: %.2f GHz)\n", wpc, ms, flops / ms / 1e9,
               (double)iters * 4 * 32 * (wpc / 4) / (ms * 1e-3) / 1e9);
    }
    return 0;
}
