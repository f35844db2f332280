// K11 / K12 - the element-wise ends of the U-NO models fused into single passes (exact-erf GELU, as F.gelu):
//
//   K11  gelu_project:  out[b][p] = bias + sum_c w[c] * gelu(pre[b][c][p])        the final projection fc2 (C -> 1) applied to
//        gelu(fc1(.)) (reference darcy_flow_uno2d.py:128-131, navier_stokes_uno2d.py:222-225, navier_stokes_uno3d.py:
//        378-381), and its backward  gpre = gelu'(pre) * w[c] * gout,  gw[c] = sum gout * gelu(pre),  gb = sum gout.
//        Stock ops: GELU (2 passes over the C-channel tensor) + a GEMM with one useful output row; backward 3 + 2 + 2 passes.
//        Here: forward reads pre once; backward reads pre once and writes gpre once.
//   K12  gelu_pad:  out[n][h][w] = gelu(s[n][h][w]) for h < H, w < W, else 0       the lift's last GELU followed by the domain
//        padding F.pad(x, [0, pad, 0, pad]) (darcy_flow_uno2d.py:103-107), and its backward gs = gelu'(s) * gy[n][h][w].
//
// Both are pure streaming kernels: 16-byte accesses, rows of consecutive pixels per wave (1 KB per instruction).
#include "uno_common.h"
#include <cstdio>

namespace uno {

__device__ __forceinline__ float gelu_f(float x) { return uno_gelu(x); }
__device__ __forceinline__ float dgelu_f(float x) { return uno_dgelu(x); }

// up to four consecutive floats row[px .. px+3] with zeros from `n` on (n = valid floats in the row, n >= 4)
// T = float, or unsigned short = bfloat16 bits (config C5: activations bf16, weights and every accumulation f32; widened on
// load, rounded to nearest even on store)
template <typename T>
__device__ __forceinline__ void load4_guard(const T* row, int px, int n, float v[4]) {
    if (px + 3 < n) {
        const float4 t = io_ld4(row + px);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = px + i < n ? io_widen(row[px + i]) : 0.f;
    }
}
template <typename T>
__device__ __forceinline__ void store4_guard(T* row, int px, int n, const float v[4]) {
    if (px + 3 < n) {
        io_store4(row + px, v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (px + i < n) io_store1(row + px + i, v[i]);
    }
}

// ------------------------------------------------------------------------------------------------ K11
constexpr int GP_MAXC = 1024;

// SPLIT: small tensors (a 64 x 64 grid at batch 32 gives 512 one-wave pixel tiles for 256 CUs, each walking all C channels): the
// four waves of a workgroup share ONE 256-pixel tile and take a quarter of the channels each; partial sums meet in LDS and are
// added in wave order (fixed -> bit-reproducible).
template <bool SPLIT, typename T>
__global__ __launch_bounds__(256) void gelu_project_fwd_kernel(const T* __restrict__ pre, const float* __restrict__ w,
                                                               const float* __restrict__ bias, T* __restrict__ out, int C, int P) {
    __shared__ float sw[GP_MAXC];
    __shared__ float sred[SPLIT ? 4 * 64 * 4 : 1];
    for (int c = threadIdx.x; c < C; c += blockDim.x) sw[c] = w[c];
    __syncthreads();
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = SPLIT ? (blockIdx.x * 64 + lane) * 4 : (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const bool live = px < P;
    if (!SPLIT && !live) return;
    const int Cs = SPLIT ? (C + 3) / 4 : C;
    const int c_lo = SPLIT ? min(wave * Cs, C) : 0, c_hi = SPLIT ? min(c_lo + Cs, C) : C;
    const T* src = pre + (size_t)b * C * P;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // channels in groups of 4, the next group's loads issued before the current one is consumed (-11 % against the
    // compiler's own unrolling, which drains each group of loads before issuing the next; the same change made the
    // backward kernel 12 % slower - it keeps the plain loop)
    auto load = [&](int c0, float v[4][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) load4_guard(src + (size_t)max(min(c0 + j, c_hi - 1), 0) * P, px, P, v[j]);
    };
    auto use = [&](int c0, const float v[4][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c0 + j < c_hi) {
                const float wc = sw[c0 + j];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fmaf(wc, gelu_f(v[j][i]), acc[i]);
            }
        }
    };
    if (live && c_lo < c_hi) {
        float va[4][4], vb[4][4];
        load(c_lo, va);
        for (int c = c_lo; c < c_hi; c += 8) {
            load(c + 4, vb);
            __builtin_amdgcn_sched_barrier(0);
            use(c, va);
            load(c + 8, va);
            __builtin_amdgcn_sched_barrier(0);
            use(c + 4, vb);
        }
    }
    const float bv = bias ? bias[0] : 0.f;
    if (SPLIT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sred[(wave * 64 + lane) * 4 + i] = acc[i];
        __syncthreads();
        if (wave != 0 || !live) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = ((sred[lane * 4 + i] + sred[(64 + lane) * 4 + i]) + sred[(128 + lane) * 4 + i]) + sred[(192 + lane) * 4 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += bv;
    store4_guard(out + (size_t)b * P, px, P, acc);
}

// one workgroup = 4 pixels per thread of one batch entry (256 threads, or 64 when the tensor is small: a 64 x 64 grid at
// batch 32 gave 128 workgroups of 256 threads for 256 CUs); partial weight / bias gradients per workgroup: part[blk][C + 1].
// Per channel every wave reduces its 256 pixels by butterfly and parks the sum in LDS; one barrier at the end, then the
// four wave sums are added in wave order (fixed order -> bit-reproducible).
template <typename T>
__global__ __launch_bounds__(256) void gelu_project_bwd_kernel(const T* __restrict__ pre, const float* __restrict__ w,
                                                               const T* __restrict__ gout, T* __restrict__ gpre,
                                                               float* __restrict__ part, int C, int P, int Cs, PixMap pm, int rev) {
    // pm: pixel window (uno_common.h) - pre, gout (one plane per batch entry) and gpre on one window; dense: pm.PS == P
    // blockIdx.z = channel split: channels [z Cs, min(C, (z + 1) Cs)); small tensors (one-wave workgroups) are split over
    // channels as well, so that the chip sees 4x the waves (a 64 x 64 grid at batch 32 gave 512 waves walking 128 channels each)
    __shared__ float sw[GP_MAXC];
    extern __shared__ float swave[];                    // [waves][C + 1]
    const int nw = blockDim.x >> 6;
    for (int c = threadIdx.x; c < C; c += blockDim.x) sw[c] = w[c];
    __syncthreads();
    const int bxs = sweep_x(rev), b = sweep_y(rev);           // (alternating sweep direction, uno_common.h: the partial-sum slot follows the work item)
    const int px = (bxs * blockDim.x + threadIdx.x) * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = px < P;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    // a windowed call has whole quads only (P = rows * cols, cols % 4 == 0): shifting the row pointer by (offset in the plane - px)
    // leaves the guards in logical pixels
    const int PS = pm.PS, fx = live ? pix_phys(pm, px) - px : 0;
    if (live) load4_guard(gout + (size_t)b * PS + fx, px, P, g);
    const T* src = pre + (size_t)b * C * PS + fx;
    T* dst = gpre + (size_t)b * C * PS + fx;
    float* mine = swave + wave * (C + 1);
    auto wave_sum = [&](float s, int slot) {
#pragma unroll
        for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) mine[slot] = s;
    };
    const int c_lo = blockIdx.z * Cs, c_hi = min(C, c_lo + Cs);
#pragma unroll 2
    for (int c = c_lo; c < c_hi; ++c) {
        float s = 0.f;
        if (live) {
            float v[4], o[4];
            load4_guard(src + (size_t)c * PS, px, P, v);
            const float wc = sw[c];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float cdf = 0.5f * (1.f + uno_erf(v[i] * 0.70710678118654752440f));
                const float pdf = 0.39894228040143267794f * __expf(-0.5f * v[i] * v[i]);
                o[i] = fmaf(v[i], pdf, cdf) * (wc * g[i]);
                s = fmaf(g[i], v[i] * cdf, s);              // g is zero past the row end
            }
            store4_guard(dst + (size_t)c * PS, px, P, o);
        }
        wave_sum(s, c);
    }
    wave_sum((g[0] + g[1]) + (g[2] + g[3]), C);
    __syncthreads();
    // part[z][blk][C + 1]: a split fills its own channels (split 0 also the bias slot C); the reduction reads exactly those
    float* prow = part + (((size_t)blockIdx.z * gridDim.y + b) * gridDim.x + bxs) * (C + 1);
    auto total = [&](int e) {
        float t = swave[e];
        for (int k = 1; k < nw; ++k) t += swave[k * (C + 1) + e];          // wave order: fixed
        return t;
    };
    for (int e = c_lo + threadIdx.x; e < c_hi; e += blockDim.x) prow[e] = total(e);
    if (blockIdx.z == 0 && threadIdx.x == 0) prow[C] = total(C);
}

__global__ __launch_bounds__(256) void gelu_project_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw,
                                                                  float* __restrict__ gb, int C, int nblk, int Cs) {
    // one wave per output entry, lanes stride over the workgroup partials of the entry's channel split, fixed order
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e > C) return;
    const int lane = threadIdx.x & 63;
    part += (size_t)(e < C ? e / Cs : 0) * nblk * (C + 1);
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += part[(size_t)k * (C + 1) + e];
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        if (e < C) gw[e] = s;
        else if (gb) gb[0] = s;
    }
}

// 256-thread workgroups unless that leaves the GPU under-filled (< 4 workgroups per CU)
static int gelu_project_threads(int B, long long P) { return (long long)B * ((P + 1023) / 1024) >= 1024 ? 256 : 64; }

int launch_gelu_project_fwd(const void* pre, const float* w, const float* bias, void* out, int B, int C, long long P, int bf16, hipStream_t s) {
    typedef unsigned short bf_t;
    if (C > GP_MAXC || P > 0x7fffffffLL || B > 65535) { set_error("gelu_project: C <= %d, pixels < 2^31, batch < 65536", GP_MAXC); return -2; }
    const int threads = gelu_project_threads(B, P);
    const bool split = threads == 64 && C >= 16;
    const unsigned nb = (unsigned)((P + 4 * threads - 1) / (4 * threads));
    {
        ProfScope prof("uno::gelu_project_fwd_kernel", (bf16 ? 2.0 : 4.0) * B * (double)P * (C + 1), s);
        if (bf16) {
            if (split) hipLaunchKernelGGL((gelu_project_fwd_kernel<true, bf_t>), dim3(nb, B), dim3(256), 0, s, (const bf_t*)pre, w, bias, (bf_t*)out, C, (int)P);
            else hipLaunchKernelGGL((gelu_project_fwd_kernel<false, bf_t>), dim3(nb, B), dim3(threads), 0, s, (const bf_t*)pre, w, bias, (bf_t*)out, C, (int)P);
        } else {
            if (split) hipLaunchKernelGGL((gelu_project_fwd_kernel<true, float>), dim3(nb, B), dim3(256), 0, s, (const float*)pre, w, bias, (float*)out, C, (int)P);
            else hipLaunchKernelGGL((gelu_project_fwd_kernel<false, float>), dim3(nb, B), dim3(threads), 0, s, (const float*)pre, w, bias, (float*)out, C, (int)P);
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("gelu_project launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

// channel splits of the backward pass: 4 for the small-tensor (one-wave workgroup) case
static int gelu_project_splits(int B, int C, long long P) { return (gelu_project_threads(B, P) == 64 && C >= 16) ? 4 : 1; }

long long gelu_project_ws_floats(int B, int C, long long P) {
    const int t = gelu_project_threads(B, P);
    return (long long)gelu_project_splits(B, C, P) * B * ((P + 4 * t - 1) / (4 * t)) * (C + 1);
}

int launch_gelu_project_bwd(const void* pre, const float* w, const void* gout, void* gpre, float* gw, float* gb, float* ws, int B,
                            int C, long long P, int bf16, hipStream_t s, const PixelWindow& win) {
    typedef unsigned short bf_t;
    if (C > GP_MAXC || P > 0x7fffffffLL || B > 65535) { set_error("gelu_project: C <= %d, pixels < 2^31, batch < 65536", GP_MAXC); return -2; }
    if (const char* why = pix_window_error(win, P)) { set_error("gelu_project: %s", why); return -2; }
    if (win.cols && (long long)C * win.plane > 0x7fffffffLL) { set_error("gelu_project: window planes too large"); return -2; }
    const PixMap pm = pix_map(win, P);
    const int threads = gelu_project_threads(B, P);
    const unsigned nb = (unsigned)((P + 4 * threads - 1) / (4 * threads));
    const int nsplit = gelu_project_splits(B, C, P), Cs = (C + nsplit - 1) / nsplit;
    const int rev = next_sweep_reversed(SWEEP_PROJ);
    {
        ProfScope prof("uno::gelu_project_bwd_kernel", (bf16 ? 2.0 : 4.0) * B * (double)P * (2 * C + 1), s);
        const size_t lds = (threads / 64) * (C + 1) * sizeof(float);
        if (bf16) hipLaunchKernelGGL(gelu_project_bwd_kernel<bf_t>, dim3(nb, B, nsplit), dim3(threads), lds, s, (const bf_t*)pre, w, (const bf_t*)gout, (bf_t*)gpre, ws, C, (int)P, Cs, pm, rev);
        else hipLaunchKernelGGL(gelu_project_bwd_kernel<float>, dim3(nb, B, nsplit), dim3(threads), lds, s, (const float*)pre, w, (const float*)gout, (float*)gpre, ws, C, (int)P, Cs, pm, rev);
    }
    hipLaunchKernelGGL(gelu_project_reduce_kernel, dim3((C + 1 + 3) / 4), dim3(256), 0, s, ws, gw, gb, C, (int)(nb * B), Cs);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("gelu_project backward launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

// ---- border of padded planes: everything outside the top-left rows x cols corner of n_planes (Hp, Wp) planes := 0 (what a
// kernel that writes only the domain window of a padded tensor leaves behind: the right strip and the bottom rows).  Four
// workgroups per plane; the bottom rows are one contiguous run, the strip 4-byte stores in runs of Wp - cols.
__global__ __launch_bounds__(256) void clear_border_kernel(float* __restrict__ t, int Hp, int Wp, int rows, int cols) {
    // blockIdx.y: quarter of the plane's work (1024 planes x 87 KB as 1024 workgroups ran at 2.3 TB/s: ~50 dependent-address stores per thread)
    float* plane = t + (size_t)blockIdx.x * Hp * Wp;
    const int part = blockIdx.y, nparts = gridDim.y;
    const int strip = Wp - cols, nstrip = rows * strip;
    for (int e = part * 256 + threadIdx.x; e < nstrip; e += 256 * nparts) {
        const int r = e / strip;
        plane[(size_t)r * Wp + cols + (e - r * strip)] = 0.f;
    }
    float* bot = plane + (size_t)rows * Wp;
    const int nbot = (Hp - rows) * Wp;
    // 16-byte stores from the first aligned element on
    const int head = min(nbot, (int)(((16 - (reinterpret_cast<uintptr_t>(bot) & 15)) & 15) >> 2));
    if (part == 0 && (int)threadIdx.x < head) bot[threadIdx.x] = 0.f;
    const int nq = (nbot - head) >> 2;
    float4* b4 = reinterpret_cast<float4*>(bot + head);
    for (int q = part * 256 + threadIdx.x; q < nq; q += 256 * nparts) b4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (part == 0)
        for (int e = head + 4 * nq + threadIdx.x; e < nbot; e += 256) bot[e] = 0.f;
}

int launch_clear_border(float* t, long long n_planes, int Hp, int Wp, int rows, int cols, hipStream_t s) {
    if (n_planes > 0x7fffffffLL || (long long)Hp * Wp > 0x7fffffffLL) { set_error("clear_border: too many planes or plane too large"); return -2; }
    if (n_planes == 0 || (rows == Hp && cols == Wp)) return 0;
    ProfScope prof("uno::clear_border_kernel", 4.0 * n_planes * ((double)Hp * Wp - (double)rows * cols), s);
    hipLaunchKernelGGL(clear_border_kernel, dim3((unsigned)n_planes, 4), dim3(256), 0, s, t, Hp, Wp, rows, cols);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("clear_border launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

// ------------------------------------------------------------------------------------------------ K12
// thread = 4 consecutive columns of one row; rows are flattened so that short rows do not leave lanes idle
template <typename T>
__global__ __launch_bounds__(256) void gelu_pad_fwd_kernel(const T* __restrict__ s, T* __restrict__ out, int H, int W, int Hp, int Wp) {
    const int n = blockIdx.y;
    const int nq = (Wp + 3) >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int h = q / nq;
    if (h >= Hp) return;
    const int w0 = (q - h * nq) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (h < H && w0 < W) {
        load4_guard(s + ((size_t)n * H + h) * W, w0, W, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = w0 + i < W ? gelu_f(v[i]) : 0.f;
    }
    store4_guard(out + ((size_t)n * Hp + h) * Wp, w0, Wp, v);
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_pad_bwd_kernel(const T* __restrict__ s, const T* __restrict__ gy, T* __restrict__ gs,
                                                           int H, int W, int Hp, int Wp) {
    const int n = blockIdx.y;
    const int nq = (W + 3) >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int h = q / nq;
    if (h >= H) return;
    const int w0 = (q - h * nq) * 4;
    float v[4], g[4], o[4];
    load4_guard(s + ((size_t)n * H + h) * W, w0, W, v);
    load4_guard(gy + ((size_t)n * Hp + h) * Wp, w0, W, g);          // only the first W columns of the padded row matter
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = dgelu_f(v[i]) * g[i];
    store4_guard(gs + ((size_t)n * H + h) * W, w0, W, o);
}

int launch_gelu_pad(const void* s, const void* gy, void* out, int n_img, int H, int W, int Hp, int Wp, int backward, int bf16, hipStream_t st) {
    typedef unsigned short bf_t;
    const double es = bf16 ? 2.0 : 4.0;
    if (n_img > 65535 || (long long)Hp * ((Wp + 3) / 4) > 0x7fffffffLL) { set_error("gelu_pad: at most 65535 images"); return -2; }
    if (!backward) {
        ProfScope prof("uno::gelu_pad_fwd_kernel", es * n_img * ((double)H * W + (double)Hp * Wp), st);
        const long long quads = (long long)Hp * ((Wp + 3) / 4);
        const dim3 grid((unsigned)((quads + 255) / 256), n_img);
        if (bf16) hipLaunchKernelGGL(gelu_pad_fwd_kernel<bf_t>, grid, dim3(256), 0, st, (const bf_t*)s, (bf_t*)out, H, W, Hp, Wp);
        else hipLaunchKernelGGL(gelu_pad_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)s, (float*)out, H, W, Hp, Wp);
    } else {
        ProfScope prof("uno::gelu_pad_bwd_kernel", es * n_img * 3.0 * H * W, st);
        const long long quads = (long long)H * ((W + 3) / 4);
        const dim3 grid((unsigned)((quads + 255) / 256), n_img);
        if (bf16) hipLaunchKernelGGL(gelu_pad_bwd_kernel<bf_t>, grid, dim3(256), 0, st, (const bf_t*)s, (const bf_t*)gy, (bf_t*)out, H, W, Hp, Wp);
        else hipLaunchKernelGGL(gelu_pad_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)s, (const float*)gy, (float*)out, H, W, Hp, Wp);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("gelu_pad launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

// ------------------------------------------------------------------------------------------------ K14
// Batched transposing copy  out[b][c][r] = in[b][r][c]  between the two layouts an activation reaches an operator block in:
// channels-first (what every kernel of this library reads and writes) and channels-last - what the reference's model files
// hand over when they `permute` the output of a channels-last nn.Linear (darcy_flow_uno2d.py:104-107: permute + F.pad gives
// conv0 an NHWC tensor) and what their autograd graph hands back.  torch's own strided copy moves the 407 MB lift output of
// the 421^2 Darcy model in 3.9 ms (0.2 TB/s: one 4-byte element per lane, 128-byte strides between lanes); this kernel
// stages 64 x 64 tiles in LDS so that both sides move whole 256-byte row segments.
// in: R rows of C contiguous floats at pitch ld_in;  out: C rows of R contiguous floats at pitch ld_out.
template <bool VIN, bool VOUT>
__global__ __launch_bounds__(256) void transpose_tile_kernel(const float* __restrict__ in, float* __restrict__ out, long long R, int C,
                                                             long long ld_in, long long sb_in, long long ld_out, long long sb_out) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const float* src = in + (size_t)blockIdx.z * sb_in;
    float* dst = out + (size_t)blockIdx.z * sb_out;
    const int q = (t & 15) * 4, rr = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rr + 16 * i;
        const long long gr = r0 + r;
        const int gc = c0 + q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (gr < R && gc < C) {
            const float* row = src + (size_t)gr * ld_in + gc;
            if (VIN && gc + 3 < C) {
                const float4 w = *reinterpret_cast<const float4*>(row);
                v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (gc + e < C) v[e] = row[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r][q + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = rr + 16 * i;
        const int gc = c0 + c;
        const long long gr = r0 + q;
        if (gc < C && gr < R) {
            float* row = dst + (size_t)gc * ld_out + gr;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tile[q + e][c];
            if (VOUT && gr + 3 < R) {
                *reinterpret_cast<float4*>(row) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (gr + e < R) row[e] = v[e];
            }
        }
    }
}

int launch_transpose_batched(const float* in, float* out, int B, long long R, int C, long long ld_in, long long sb_in, long long ld_out,
                             long long sb_out, hipStream_t st) {
    if (B == 0 || R == 0 || C == 0) return 0;
    const long long tr = (R + 63) / 64;
    const int tc = (C + 63) / 64;
    if (B > 65535 || tc > 65535 || tr > 0x7fffffffLL) { set_error("transpose_batched: at most 65535 batch entries and 4 M columns"); return -2; }
    ProfScope prof("uno::transpose_tile_kernel", 8.0 * B * (double)R * C, st);
    // 16-byte accesses where every row starts on a 16-byte boundary
    const bool vin = (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (ld_in & 3) == 0 && (sb_in & 3) == 0;
    const bool vout = (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (ld_out & 3) == 0 && (sb_out & 3) == 0;
    const dim3 grid((unsigned)tr, (unsigned)tc, (unsigned)B);
#define UNO_TR_LAUNCH(A_, B_) hipLaunchKernelGGL((transpose_tile_kernel<A_, B_>), grid, dim3(256), 0, st, in, out, R, C, ld_in, sb_in, ld_out, sb_out)
    if (vin && vout) UNO_TR_LAUNCH(true, true);
    else if (vin) UNO_TR_LAUNCH(true, false);
    else if (vout) UNO_TR_LAUNCH(false, true);
    else UNO_TR_LAUNCH(false, false);
#undef UNO_TR_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("transpose_batched launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
