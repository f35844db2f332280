// Micro-probe: how close to the f32 MFMA issue rate (32 cycles per v_mfma_f32_16x16x4_f32 per SIMD) do
// inner-loop structures of the DFT kernels get?  Variants add VALU / LDS work per MFMA group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ unsigned wrap_add(unsigned i, unsigned inc, unsigned lim) { unsigned t = i + inc; return min(t, t - lim); }

template <int V>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, int iters, int W) {
    __shared__ float2 tab[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) tab[i] = make_float2(in[i & 255], in[(i + 7) & 255]);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x0 = in[lane], x1 = in[lane + 64], x2 = in[lane + 128], x3 = in[lane + 192];
    unsigned idx0 = 8u * ((lane * 5) % W), idx1 = 8u * ((lane * 11) % W);
    const unsigned st0 = 8u * (lane & 15), st1 = 8u * ((lane & 15) + 16), W8 = 8u * W;
    float2 t0 = tab[0], t1 = tab[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float E = x0, D = x1;
            if (V >= 1) { E = x0 + x3; D = x0 - x3; x0 = x1; x1 = x2; x2 = x3; x3 = E * 0.5f; }
            float2 n0 = t0, n1 = t1;
            if (V >= 2) {
                n0 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(tab) + idx0);
                n1 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(tab) + idx1);
                if (V == 2 || V == 4) { idx0 = wrap_add(idx0, st0, W8); idx1 = wrap_add(idx1, st1, W8); }
                else { idx0 = (idx0 + 512) & 8191; idx1 = (idx1 + 512) & 8191; }   // V==3: conflict-free-ish cheap walk
            }
            a0 = mfma16(E, t0.x, a0);
            a1 = mfma16(D, t0.y, a1);
            a2 = mfma16(E, t1.x, a2);
            a3 = mfma16(D, t1.y, a3);
            t0 = n0; t1 = n1;
        }
    }
    f32x4 r = a0 + a1 + a2 + a3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + x0;
}

template <int V>
void run(const char* name, float* in, float* out, int wavesPerSimd) {
    const int iters = 2000;
    const int blocks = 256 * wavesPerSimd;       // 256 CUs, blocks of 4 waves (1 per SIMD)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, in, out, iters, 421);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, in, out, iters, 421);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * wavesPerSimd;
    const double ns_per_mfma = ms * 1e6 / mfma_per_simd;
    printf("%-28s waves/SIMD %d: %7.1f us  %5.1f ns/MFMA/SIMD  (= %4.1f cyc @2.1GHz; ideal 32)  util %.0f%%\n", name, wavesPerSimd,
           ms * 1e3, ns_per_mfma, ns_per_mfma * 2.1, 100.0 * 32.0 / (ns_per_mfma * 2.1));
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 8 * 256);
    std::vector<float> h(1024, 0.5f); for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; ++w) {
        run<0>("mfma only", in, out, w);
        run<1>("+ E/D valu", in, out, w);
        run<3>("+ lds b64 (cheap idx)", in, out, w);
        run<2>("+ lds b64 + wrap walk", in, out, w);
    }
    return 0;
}
