// peak rate of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products = 512 flop) against v_mfma_f32_16x16x4_f32 (2048 flop):
// waves x iterations of 16 independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = KIND ? __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void run(float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        const int iters = 10000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 2.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 16 * (KIND ? 512 : 2048);
        printf("%s blocks %d: %.3f ms  %.1f TFLOP/s  (cycles per MFMA per SIMD at 2.4 GHz: %.1f)\n", KIND ? "4x4x1_16b " : "16x16x4   ", blocks, ms,
               flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)blocks * 4 * iters * 16 / 1024.0));
    }
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    run<0>(out);
    run<1>(out);
    return 0;
}
