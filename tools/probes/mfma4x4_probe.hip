// Layout probe for v_mfma_f32_4x4x1_16b_f32: which lanes/registers hold A[i], B[j], D[i][j] of block b?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int i = 0; i < 64; ++i) { ha[i] = 1.0f + i; hb[i] = 100.0f + 3 * i; }
    float *a, *b, *d; hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    // H1: D[lane][r] = A[4*(lane/4) + r] * B[lane]
    int ok1 = 1, ok2 = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        if (std::fabs(hd[l * 4 + r] - ha[4 * (l / 4) + r] * hb[l]) > 1e-3) ok1 = 0;
        if (std::fabs(hd[l * 4 + r] - ha[l] * hb[4 * (l / 4) + r]) > 1e-3) ok2 = 0;
    }
    printf("H1 (D[lane][r] = A[4*(lane/4)+r] * B[lane]): %s\nH2 (D[lane][r] = A[lane] * B[4*(lane/4)+r]): %s\n", ok1 ? "YES" : "no", ok2 ? "YES" : "no");
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g   (a=%g b=%g)\n", l, hd[l*4], hd[l*4+1], hd[l*4+2], hd[l*4+3], ha[l], hb[l]);
    return 0;
}
