// Why does K3 (pruned inverse DFT) write at 2.6 TB/s when a fill kernel writes at 6.8 TB/s?  This probe reproduces K3's
// store structure without its arithmetic: 1024 images of H x W floats, a workgroup of NWT waves = G images x NW waves, each
// wave owns the 16-row tiles rt = wsub, wsub + NW, ... of its image and sweeps them in 64-column chunks from the left edge
// and (mirrored) from the right edge.  Variants of the store shape / alignment / instruction width are timed against each
// other, with an optional block of dependent MFMAs between chunks (the compute a real wave does there).
//   hipcc -O3 --offload-arch=gfx950 store_probe.hip -o store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

enum Mode {
    K3_PATTERN = 0,     // 4 rows x 256 B per instruction, 16 B per lane at 4-byte alignment, left + mirrored right
    ALIGNED16 = 1,      // same, lane addresses rounded down to 16 B (aligned dwordx4)
    DWORD = 2,          // 1 row x 256 B per instruction with dword stores (64 lanes x 4 B), 4x the instructions
    ROW1K = 3,          // 1 row x 1 KB per instruction (256 columns per chunk), 16 B per lane, 4-byte alignment
    ROW512 = 4,         // 2 rows x 512 B per instruction (128 columns per chunk)
    TILE_CONTIG = 5,    // the 16-row tile as one contiguous run: 1 KB per instruction, 16-byte aligned
    LEFT_ONLY = 6,      // K3 pattern, left half only (half the bytes)
    ALIGNED64 = 7,      // row segments snapped to 64-byte boundaries
};

template <int MODE>
__global__ __launch_bounds__(768) void wr(float* out, int H, int W, int NW, int work, float seed) {
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = tid >> 6, NWT = blockDim.x >> 6;
    const int slot = wave / NW, wsub = wave - slot * NW;
    const int image = blockIdx.x * (NWT / NW) + slot;
    float* img = out + (size_t)image * H * W;
    const int nrt = H / 16;                      // full tiles only
    const int Wh = W >> 1;
    f32x4 acc = {seed, 1.f, 2.f, 3.f};
    for (int rt = wsub; rt < nrt; rt += NW) {
        float* tile = img + (size_t)rt * 16 * W;
        if (MODE == TILE_CONTIG) {
            const size_t a0 = ((size_t)(tile - out) + 3) & ~(size_t)3;
            const int n = 16 * W / 256;
            for (int i = 0; i < n; ++i) {
                for (int k = 0; k < work / 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[0], acc[1], acc, 0, 0, 0);
                *reinterpret_cast<f32x4*>(out + a0 + (size_t)i * 256 + 4 * lane) = acc;
            }
            continue;
        }
        const int CH = MODE == ROW1K ? 256 : (MODE == ROW512 ? 128 : 64);
        for (int c0 = 0; c0 + CH - 1 <= Wh; c0 += CH) {
            for (int k = 0; k < work * (CH / 64); ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[0], acc[1], acc, 0, 0, 0);
            const int cr0 = W - c0 - (CH - 1);
            if (MODE == K3_PATTERN || MODE == ALIGNED16 || MODE == LEFT_ONLY || MODE == ALIGNED64) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float* rowp = tile + (size_t)(4 * q + kk) * W;
                    float* pl = rowp + c0 + 4 * r16;
                    float* pr = rowp + cr0 + 4 * r16;
                    if (MODE == ALIGNED16) { pl = (float*)((uintptr_t)pl & ~(uintptr_t)15); pr = (float*)((uintptr_t)pr & ~(uintptr_t)15); }
                    if (MODE == ALIGNED64) {
                        pl = (float*)(((uintptr_t)(rowp + c0) & ~(uintptr_t)63) + 16 * r16);
                        pr = (float*)(((uintptr_t)(rowp + cr0) & ~(uintptr_t)63) + 16 * r16);
                    }
                    *reinterpret_cast<f4u*>(pl) = f4u{{acc[0], acc[1], acc[2], acc[3]}};
                    if (MODE != LEFT_ONLY) *reinterpret_cast<f4u*>(pr) = f4u{{acc[3], acc[2], acc[1], acc[0]}};
                }
            } else if (MODE == DWORD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* rowp = tile + (size_t)r * W;
                    rowp[c0 + lane] = acc[r & 3];
                    rowp[cr0 + lane] = acc[(r + 1) & 3];
                }
            } else if (MODE == ROW1K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* rowp = tile + (size_t)r * W;
                    *reinterpret_cast<f4u*>(rowp + c0 + 4 * lane) = f4u{{acc[0], acc[1], acc[2], acc[3]}};
                    *reinterpret_cast<f4u*>(rowp + cr0 + 4 * lane) = f4u{{acc[3], acc[2], acc[1], acc[0]}};
                }
            } else if (MODE == ROW512) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float* rowp = tile + (size_t)(2 * q + (lane >> 5)) * W;
                    *reinterpret_cast<f4u*>(rowp + c0 + 4 * (lane & 31)) = f4u{{acc[0], acc[1], acc[2], acc[3]}};
                    *reinterpret_cast<f4u*>(rowp + cr0 + 4 * (lane & 31)) = f4u{{acc[3], acc[2], acc[1], acc[0]}};
                }
            }
        }
    }
}

template <int MODE>
static float run(float* const* bufs, int nimg, int H, int W, int NW, int G, int work) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(wr<MODE>, dim3(nimg / G), dim3(64 * NW * G), 0, 0, bufs[it & 3], H, W, NW, work, (float)it);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main(int argc, char** argv) {
    const int nimg = 1024, H = 416;
    float* bufs[4];
    for (int i = 0; i < 4; ++i) { hipMalloc(&bufs[i], (size_t)nimg * 421 * 448 * 4 + 4096); hipMemset(bufs[i], 0, (size_t)nimg * 421 * 448 * 4); }
    const char* names[] = {"k3 pattern", "aligned16", "dword", "row 1KB", "row 512B", "tile contig", "left only", "aligned64"};
    for (int W : {421, 448}) {
        for (int work : {0, 40}) {
            for (int geo = 0; geo < 3; ++geo) {
                const int NW = 3, G = geo == 0 ? 4 : (geo == 1 ? 2 : 1);
                printf("W=%d work=%d MFMAs/chunk, %d waves per workgroup (G=%d)\n", W, work, NW * G, G);
                float t[8];
                t[0] = run<0>(bufs, nimg, H, W, NW, G, work); t[1] = run<1>(bufs, nimg, H, W, NW, G, work);
                t[2] = run<2>(bufs, nimg, H, W, NW, G, work); t[3] = run<3>(bufs, nimg, H, W, NW, G, work);
                t[4] = run<4>(bufs, nimg, H, W, NW, G, work); t[5] = run<5>(bufs, nimg, H, W, NW, G, work);
                t[6] = run<6>(bufs, nimg, H, W, NW, G, work); t[7] = run<7>(bufs, nimg, H, W, NW, G, work);
                for (int m = 0; m < 8; ++m) {
                    // bytes actually covered: chunks of CH columns from both edges while c0 + CH - 1 <= Wh
                    const int CH = m == 3 ? 256 : (m == 4 ? 128 : 64);
                    int nch = 0; for (int c0 = 0; c0 + CH - 1 <= (W >> 1); c0 += CH) ++nch;
                    double bytes = (double)nimg * H * (double)nch * CH * 4 * (m == 6 ? 1 : 2);
                    if (m == 5) bytes = (double)nimg * (H / 16) * (16 * W / 256) * 1024.0;
                    printf("   %-12s %7.1f us  %5.2f TB/s\n", names[m], t[m], bytes / t[m] / 1e6);
                }
            }
        }
    }
    return 0;
}
