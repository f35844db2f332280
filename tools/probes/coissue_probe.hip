// Do VALU instructions of one wave overlap the f32 / bf16 MFMAs of ANOTHER wave on the same SIMD?  And inside one wave?
// Workgroup of 8 waves on one CU (wave w and w + 4 share SIMD w % 4).  Roles by mode:
//   0: waves 0-3 MFMA only (4-7 exit)     1: waves 4-7 VALU only      2: 0-3 MFMA, 4-7 VALU (cross-wave overlap)
//   3: every wave MFMA then VALU in one loop body, compiler's order      4: MFMA only, all 8 waves      5: VALU only, all 8 waves
// prints cycles (s_memtime) per loop iteration: NM MFMAs and NV VALU FMAs per iteration and wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NM = 8, NV = 64;
template <int KIND>      // 0: v_mfma_f32_16x16x4_f32   1: v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int mode, float a, float b) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = mode == 0 ? wave < 4 : mode == 1 ? false : mode == 2 ? wave < 4 : mode == 3 ? true : mode == 4;
    const bool do_v = mode == 0 ? false : mode == 1 ? wave >= 4 : mode == 2 ? wave >= 4 : mode == 3 ? true : mode == 5;
    f32x4 acc[NM];
    for (int i = 0; i < NM; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a * i + threadIdx.x;
    const float av = a + threadIdx.x, bv = b;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a + i); bh[i] = (__bf16)(b + i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) v[(i * (NV / NM) + j) & 15] = fmaf(v[(i * (NV / NM) + j) & 15], a, b);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
            }
        }
    } else if (do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 15] = fmaf(v[j & 15], a, b);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = (do_m || do_v) ? t1 - t0 : 0;
}
template <int KIND>
static void run(const char* name) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 8 * 8);
    unsigned long long h[256 * 8];
    const int iters = 2000;
    const char* modes[] = {"MFMA waves 0-3 alone", "VALU waves 4-7 alone", "MFMA waves 0-3 + VALU waves 4-7", "MFMA + VALU in every wave (8 waves)", "MFMA in all 8 waves", "VALU in all 8 waves"};
    printf("%s: per iteration and wave %d MFMAs, %d VALU FMAs\n", name, NM, NV);
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, cyc, iters, mode, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double sm = 0, sv = 0; int nm = 0, nv = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) {
                if (!h[b * 8 + w]) continue;
                const bool is_m = mode == 0 || mode == 3 || mode == 4 || (mode == 2 && w < 4);
                if (is_m) { sm += h[b * 8 + w]; ++nm; } else { sv += h[b * 8 + w]; ++nv; }
            }
        printf("   %-40s", modes[mode]);
        if (nm) printf("  MFMA-role waves: %.0f cycles / iteration", sm / nm / iters);
        if (nv) printf("  VALU-role waves: %.0f cycles / iteration", sv / nv / iters);
        printf("\n");
    }
}
int main() {
    run<0>("v_mfma_f32_16x16x4_f32");
    run<1>("v_mfma_f32_16x16x32_bf16");
    return 0;
}
