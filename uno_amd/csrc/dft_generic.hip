// Any-mode-count forms of the pruned transforms (K1g / K3g / K5g / K6g).
//
// The MFMA kernels are compiled for modes1 <= 40 and modes2 <= 48 - every layer of every reference model.  The reference's
// DEFAULT mode counts are larger (SpectralConv2d_Uno: modes1 = dim1//2 - 1, modes2 = dim2//2, integral_operators.py:153-158;
// SpectralConv3d_Uno: (dim1, dim2, dim3//2 + 1), :331-333), so that a layer built without explicit modes must still run:
// these kernels take any 1 <= modes1 <= rows, modes2 <= cols/2 + 1.  They are plain f32 FMA loops (two passes through a
// workspace, one DFT axis per pass) - a correctness path for shapes no reference model uses, not a tuned one.
#include "uno_common.h"
#include <cstdio>

namespace uno {

// pass 1 of the forward transform: T[i][h][l] = sum_w x[i][h][w] exp(-2 pi i l w / W), l < m2; one workgroup per image row
__global__ __launch_bounds__(256) void gen_rows_fwd_kernel(const float* __restrict__ x, float2* __restrict__ T, const float2* __restrict__ twW,
                                                           int W, int m2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sx = reinterpret_cast<float*>(smem);             // [W]
    float2* stw = reinterpret_cast<float2*>(sx + ((W + 1) & ~1));      // [W]
    const float* row = x + (size_t)blockIdx.x * W;
    for (int w = threadIdx.x; w < W; w += blockDim.x) { sx[w] = row[w]; stw[w] = twW[w]; }
    __syncthreads();
    for (int l = threadIdx.x; l < m2; l += blockDim.x) {
        float ar = 0.f, ai = 0.f;
        int idx = 0;
        for (int w = 0; w < W; ++w) {
            const float2 t = stw[idx];
            ar = fmaf(sx[w], t.x, ar);
            ai = fmaf(-sx[w], t.y, ai);
            idx += l; if (idx >= W) idx -= W;
        }
        T[(size_t)blockIdx.x * m2 + l] = make_float2(ar, ai);
    }
}

// pass 2: X[i][j][l] = f * sum_h T[i][h][l] exp(-2 pi i K_j h / H); one workgroup per (image, corner row j)
__global__ __launch_bounds__(256) void gen_cols_fwd_kernel(const float2* __restrict__ T, Dft2dParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* stw = reinterpret_cast<float2*>(smem);           // [H]
    const int H = p.H, m1 = p.m1, m2 = p.m2;
    const int img = blockIdx.x / (2 * m1), j = blockIdx.x - img * 2 * m1;
    for (int h = threadIdx.x; h < H; h += blockDim.x) stw[h] = p.twH[h];
    __syncthreads();
    const int K = p.rowfreq ? p.rowfreq[j] : corner_freq(j, m1, H);
    const float2* Ti = T + (size_t)img * H * m2;
    float2* out = reinterpret_cast<float2*>(p.out) + (spectrum_index(p, img) * 2 * m1 + j) * m2;
    const float keep = (p.mask && !p.rowfreq && !row_survives(j, m1, H)) ? 0.f : p.scale;
    for (int l = threadIdx.x; l < m2; l += blockDim.x) {
        float ar = 0.f, ai = 0.f;
        int idx = 0;
        for (int h = 0; h < H; ++h) {
            const float2 t = stw[idx], v = Ti[(size_t)h * m2 + l];
            // (vr + i vi) (c - i s)
            ar = fmaf(v.x, t.x, fmaf(v.y, t.y, ar));
            ai = fmaf(v.y, t.x, fmaf(-v.x, t.y, ai));
            idx += K; if (idx >= H) idx -= H;
        }
        const float f = keep * (p.herm ? herm_weight(l, p.W) : 1.0f);
        out[l] = make_float2(ar * f, ai * f);
    }
}

// inverse pass 1: U[i][h][l] = sum_j f(j, l) O[i][j][l] exp(+2 pi i K_j h / H); one workgroup per (image, output row h)
__global__ __launch_bounds__(256) void gen_cols_inv_kernel(float2* __restrict__ U, Dft2dParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* stw = reinterpret_cast<float2*>(smem);           // [H]
    const int H = p.H, m1 = p.m1, m2 = p.m2;
    const int img = blockIdx.x / H, h = blockIdx.x - img * H;
    for (int n = threadIdx.x; n < H; n += blockDim.x) stw[n] = p.twH[n];
    __syncthreads();
    const float2* O = reinterpret_cast<const float2*>(p.in) + spectrum_index(p, img) * 2 * m1 * m2;
    for (int l = threadIdx.x; l < m2; l += blockDim.x) {
        float ar = 0.f, ai = 0.f;
        for (int j = 0; j < 2 * m1; ++j) {
            if (p.mask && !p.rowfreq && !row_survives(j, m1, H)) continue;
            const int K = p.rowfreq ? p.rowfreq[j] : corner_freq(j, m1, H);
            const float2 t = stw[(int)(((long long)K * h) % H)], v = O[(size_t)j * m2 + l];
            // (vr + i vi) (c + i s)
            ar = fmaf(v.x, t.x, fmaf(-v.y, t.y, ar));
            ai = fmaf(v.y, t.x, fmaf(v.x, t.y, ai));
        }
        const float f = p.scale * (p.herm ? herm_weight(l, p.W) : 1.0f);
        U[((size_t)img * H + h) * m2 + l] = make_float2(ar * f, ai * f);
    }
}

// inverse pass 2: y[i][h][w] = Re sum_l U[i][h][l] exp(+2 pi i l w / W); one workgroup per image row
__global__ __launch_bounds__(256) void gen_rows_inv_kernel(const float2* __restrict__ U, float* __restrict__ y, const float2* __restrict__ twW,
                                                           int W, int m2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* su = reinterpret_cast<float2*>(smem);            // [m2]
    float2* stw = su + m2;                                   // [W]
    for (int l = threadIdx.x; l < m2; l += blockDim.x) su[l] = U[(size_t)blockIdx.x * m2 + l];
    for (int w = threadIdx.x; w < W; w += blockDim.x) stw[w] = twW[w];
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
        float acc = 0.f;
        int idx = 0;
        for (int l = 0; l < m2; ++l) {
            const float2 t = stw[idx], v = su[l];
            acc = fmaf(v.x, t.x, fmaf(-v.y, t.y, acc));
            idx += w; if (idx >= W) idx -= W;
        }
        y[(size_t)blockIdx.x * W + w] = acc;
    }
}

static int check_launch(const char* who) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s launch: %s", who, hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_dft2d_generic(const Dft2dParams& p, bool inverse, void* ws_, size_t ws_bytes, hipStream_t s) {
    if (p.bf16) { set_error("dft2d (any-mode form): bfloat16 images need modes1 <= 40 and modes2 <= 48"); return -2; }
    const size_t lds_rows = (size_t)((p.W + 1) & ~1) * 4 + (size_t)p.W * 8 + (size_t)p.m2 * 8, lds_cols = (size_t)p.H * 8;
    if (lds_rows > 64 * 1024 || lds_cols > 64 * 1024) { set_error("dft2d (any-mode form): grid %dx%d too large", p.H, p.W); return -3; }
    // the intermediate (n_img, H, m2) spectrum lives in the CALLER's scratch (uno_scratch_provide): the library allocates nothing
    const size_t bytes = (size_t)p.n_img * p.H * p.m2 * sizeof(float2);
    if (!ws_ || ws_bytes < bytes) {
        set_error("dft2d (any-mode form, modes %d x %d beyond the MFMA kernels' range): needs %zu bytes of scratch, %zu provided - register a "
                  "device buffer with uno_scratch_provide (uno_dft2d_any_ws_bytes)", p.m1, p.m2, bytes, ws_ ? ws_bytes : (size_t)0);
        return -6;
    }
    float2* ws = static_cast<float2*>(ws_);
    int rc = 0;
    {
        ProfScope prof(inverse ? "uno::dft2d_inv_generic" : "uno::dft2d_fwd_generic",
                       (double)p.n_img * ((double)p.H * p.W * 4.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        if (!inverse) {
            hipLaunchKernelGGL(gen_rows_fwd_kernel, dim3(p.n_img * p.H), dim3(256), lds_rows, s, p.in, ws, p.twW, p.W, p.m2);
            hipLaunchKernelGGL(gen_cols_fwd_kernel, dim3(p.n_img * 2 * p.m1), dim3(256), lds_cols, s, ws, p);
        } else {
            hipLaunchKernelGGL(gen_cols_inv_kernel, dim3(p.n_img * p.H), dim3(256), lds_cols, s, ws, p);
            hipLaunchKernelGGL(gen_rows_inv_kernel, dim3(p.n_img * p.H), dim3(256), lds_rows, s, ws, p.out, p.twW, p.W, p.m2);
        }
        rc = check_launch("dft2d (any-mode form)");
    }
    return rc;
}

// ---- complex pruned DFT along the leading axis of (n_img, H, C) <-> corner-major (n_img, 4, m1, m2, m3), any modes1
__device__ __forceinline__ long long gen_corner_major(int j, int c, int m1, int m2, int m3) {
    const int j2 = c / m3, n = c - j2 * m3;
    const int rc = j >= m1, cc = j2 >= m2;
    return (((long long)(rc + 2 * cc) * m1 + (j - rc * m1)) * m2 + (j2 - cc * m2)) * m3 + n;
}

__global__ __launch_bounds__(256) void gen_cdft_kernel(CdftParams p, int inverse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* stw = reinterpret_cast<float2*>(smem);           // [H]
    const int H = p.H, C = p.C, m1 = p.m1;
    for (int n = threadIdx.x; n < H; n += blockDim.x) stw[n] = p.tw[n];
    __syncthreads();
    if (!inverse) {
        // one workgroup per (volume, corner row j): X[j][c] = f sum_h Z[h][c] exp(-i theta_j h)
        const int img = blockIdx.x / (2 * m1), j = blockIdx.x - img * 2 * m1;
        const int K = p.rowfreq ? p.rowfreq[j] : corner_freq(j, m1, H);
        const float f = (p.mask && !p.rowfreq && !row_survives(j, m1, H)) ? 0.f : p.scale;
        const float2* in = reinterpret_cast<const float2*>(p.in) + (size_t)img * H * C;
        float2* out = reinterpret_cast<float2*>(p.out) + (size_t)img * 4 * m1 * p.m2 * p.m3;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float ar = 0.f, ai = 0.f;
            int idx = 0;
            for (int h = 0; h < H; ++h) {
                const float2 t = stw[idx], v = in[(size_t)h * C + c];
                ar = fmaf(v.x, t.x, fmaf(v.y, t.y, ar));
                ai = fmaf(v.y, t.x, fmaf(-v.x, t.y, ai));
                idx += K; if (idx >= H) idx -= H;
            }
            out[gen_corner_major(j, c, m1, p.m2, p.m3)] = make_float2(ar * f, ai * f);
        }
    } else {
        // one workgroup per (volume, plane h): Z[h][c] = sum_j keep_j O[j][c] exp(+i theta_j h)
        const int img = blockIdx.x / H, h = blockIdx.x - img * H;
        const float2* in = reinterpret_cast<const float2*>(p.in) + (size_t)img * 4 * m1 * p.m2 * p.m3;
        float2* out = reinterpret_cast<float2*>(p.out) + ((size_t)img * H + h) * C;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float ar = 0.f, ai = 0.f;
            for (int j = 0; j < 2 * m1; ++j) {
                if (p.mask && !p.rowfreq && !row_survives(j, m1, H)) continue;
                const int K = p.rowfreq ? p.rowfreq[j] : corner_freq(j, m1, H);
                const float2 t = stw[(int)(((long long)K * h) % H)], v = in[gen_corner_major(j, c, m1, p.m2, p.m3)];
                ar = fmaf(v.x, t.x, fmaf(-v.y, t.y, ar));
                ai = fmaf(v.y, t.x, fmaf(v.x, t.y, ai));
            }
            out[c] = make_float2(ar * p.scale, ai * p.scale);
        }
    }
}

int launch_cdft_generic(const CdftParams& p, bool inverse, hipStream_t s) {
    const size_t lds = (size_t)p.H * sizeof(float2);
    if (lds > 64 * 1024) { set_error("cdft (any-mode form): axis length %d too large", p.H); return -3; }
    {
        ProfScope prof(inverse ? "uno::cdft_inv_generic" : "uno::cdft_fwd_generic",
                       (double)p.n_img * 8.0 * ((double)p.H * p.C + 4.0 * p.m1 * p.m2 * p.m3), s);
        hipLaunchKernelGGL(gen_cdft_kernel, dim3(inverse ? p.n_img * p.H : p.n_img * 2 * p.m1), dim3(256), lds, s, p, inverse ? 1 : 0);
    }
    return check_launch("cdft (any-mode form)");
}

}  // namespace uno
