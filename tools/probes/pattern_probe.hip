// How does HBM bandwidth depend on the access shape of the DFT kernels?  Each wave owns a tile of 16 image rows
// (row length W floats, rows contiguous) and sweeps it in column pieces of P floats per row: per step it touches
// 16 separate P*4-byte segments.  P = W means "one contiguous tile".  Measured for loads and for stores.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void rd(const float* in, float* out, int H, int W, int P, int nimg) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int tiles = (H + 15) / 16;
    if (wave >= nimg * tiles) return;
    const int img = wave / tiles, rt = wave % tiles;
    const float* base = in + (size_t)img * H * W + (size_t)rt * 16 * W;
    const int rows = min(16, H - rt * 16);
    float acc = 0.f;
    // lanes: P floats per row -> lanes_per_row = P/4 (float4 each); rows_per_instr = 64 / lanes_per_row
    const int lpr = P / 4, rpi = 64 / lpr;
    for (int c0 = 0; c0 + P <= W; c0 += P)
        for (int r0 = 0; r0 < rows; r0 += rpi) {
            const int r = r0 + lane / lpr;
            if (r < rows) {
                const float4 v = *reinterpret_cast<const float4*>(base + (size_t)r * W + c0 + 4 * (lane % lpr));
                acc += v.x + v.y + v.z + v.w;
            }
        }
    if (acc == 12345.f) out[wave] = acc;
}
__global__ __launch_bounds__(256) void wr(float* out, int H, int W, int P, int nimg) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int tiles = (H + 15) / 16;
    if (wave >= nimg * tiles) return;
    const int img = wave / tiles, rt = wave % tiles;
    float* base = out + (size_t)img * H * W + (size_t)rt * 16 * W;
    const int rows = min(16, H - rt * 16);
    const int lpr = P / 4, rpi = 64 / lpr;
    for (int c0 = 0; c0 + P <= W; c0 += P)
        for (int r0 = 0; r0 < rows; r0 += rpi) {
            const int r = r0 + lane / lpr;
            if (r < rows) *reinterpret_cast<float4*>(base + (size_t)r * W + c0 + 4 * (lane % lpr)) = make_float4(1.f, 2.f, 3.f, (float)lane);
        }
}
int main() {
    const int nimg = 1024, H = 416, W = 448;        // aligned stand-in for 421 x 421 (row = 1792 B)
    // four buffers used round-robin: 3 GB >> 256 MB Infinity Cache, so every pass is cold
    float *bufs[4], *b; hipMalloc(&b, 1 << 20);
    for (int i = 0; i < 4; ++i) { hipMalloc(&bufs[i], (size_t)nimg * H * W * 4); hipMemset(bufs[i], 0, (size_t)nimg * H * W * 4); }
    const int waves = nimg * ((H + 15) / 16);
    const int blocks = (waves * 64 + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int P : {16, 32, 64, 128, 256}) {
        for (int pass = 0; pass < 2; ++pass) {
            float best = 1e9;
            for (int it = 0; it < 8; ++it) {
                float* a = bufs[it & 3];
                hipEventRecord(e0);
                if (pass == 0) hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, a, b, H, W, P, nimg);
                else hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, a, H, W, P, nimg);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (double)nimg * H * (W / P * P) * 4;
            printf("%s piece %4d B x 16 rows: %7.1f us  %5.2f TB/s\n", pass ? "store" : "load ", P * 4, best * 1e3, bytes / best / 1e9);
        }
    }
    return 0;
}
