// K5 / K6 - pruned complex DFT along the leading (row) axis of a 3-D spectral convolution.
//
// SpectralConv3d_Uno (reference integral_operators.py:385-427) transforms three axes (H, W, T).  The
// trailing two are handled plane by plane with the 2-D kernels (K1 / K3: W carries corner rows, T is
// the half-spectrum axis); these two kernels add the remaining full-complex axis H:
//
//   K5  X[img][corner][j1][j2][n] = keep_j * scale * sum_h e^{-2 pi i K_j h / H} Z[img][h][j2'][n]
//   K6  Z[img][h][j2'][n]         = sum_j keep_j e^{+2 pi i K_j h / H} O[img][corner][j1][j2][n]
//
// where Z is the per-plane truncated spectrum (H planes of 2*m2 x m3 complex), j runs over the 2*m1
// corner rows of H and the truncated 3-D spectrum is stored corner-major
// (corner = (j >= m1) + 2 * (j2' >= m2), i.e. weights1..4 order) so that every weight tensor faces one
// contiguous run of m1*m2*m3 modes in the per-mode GEMM (K2).
//
// Both are M x 16-column x K complex GEMMs against twiddles on v_mfma_f32_16x16x4_f32: one wave per
// 16-column tile, the twiddle (A operand) gathered from an LDS table by an integer-walked phase index,
// the data (B operand) read as 128-byte runs of complex64.
#include "uno_common.h"
#include <cstdio>

namespace uno {

__device__ __forceinline__ long long corner_major_offset(int j, int c, int m1, int m2, int m3) {
    // j in [0, 2 m1), c = j2 * m3 + n with j2 in [0, 2 m2)
    const int j2 = c / m3, n = c - j2 * m3;
    const int rc = j >= m1, cc = j2 >= m2;
    const int jj1 = j - rc * m1, jj2 = j2 - cc * m2;
    return (((long long)(rc + 2 * cc) * m1 + jj1) * m2 + jj2) * m3 + n;
}

// K5: in (n_img, H, C) c64 -> out (n_img, 4, m1, m2, m3) c64, C = 2*m2*m3
template <int MT>
__global__ __launch_bounds__(256) void cdft_fwd_kernel(CdftParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* sTw = reinterpret_cast<float2*>(smem);
    const int H = p.H, C = p.C, m1 = p.m1;
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int n = tid; n < H; n += blockDim.x) sTw[n] = p.tw[n];
    __syncthreads();
    const int ctile = blockIdx.y * 4 + wave;
    if (ctile * 16 >= C) return;
    const int c = ctile * 16 + r16;
    const bool cvalid = c < C;
    const unsigned H8 = 8u * H;
    const float2* in = reinterpret_cast<const float2*>(p.in) + (size_t)blockIdx.x * H * C;

    unsigned idx[MT], step4[MT];
    bool jvalid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int j = 16 * mt + r16;
        jvalid[mt] = j < 2 * m1;
        const int K = jvalid[mt] ? (p.rowfreq ? p.rowfreq[j] : corner_freq(j, m1, H)) : 0;
        idx[mt] = 8u * (unsigned)(((long long)K * kk) % H);
        step4[mt] = 8u * (unsigned)(((long long)4 * K) % H);
    }
    f32x4 Xr[MT], Xi[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { Xr[mt] = f32x4{0, 0, 0, 0}; Xi[mt] = f32x4{0, 0, 0, 0}; }

    for (int k0 = 0; k0 < H; k0 += 4) {
        const int h = k0 + kk;
        float2 v = make_float2(0.f, 0.f);
        if (h < H && cvalid) v = in[(size_t)h * C + c];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float2 tw = lds_tw(sTw, idx[mt]);
            const bool ok = jvalid[mt] && h < H;
            const float ac = ok ? tw.x : 0.f, as = ok ? tw.y : 0.f;
            // e^{-i theta} (vr + i vi) = (c vr + s vi) + i (c vi - s vr)
            Xr[mt] = mfma16(ac, v.x, Xr[mt]);
            Xi[mt] = mfma16(ac, v.y, Xi[mt]);
            Xr[mt] = mfma16(as, v.y, Xr[mt]);
            Xi[mt] = mfma16(-as, v.x, Xi[mt]);
            idx[mt] = wrap_add(idx[mt], step4[mt], H8);
        }
    }
    if (!cvalid) return;
    float2* out = reinterpret_cast<float2*>(p.out) + (size_t)blockIdx.x * 4 * m1 * p.m2 * p.m3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * mt + 4 * kk + r;
            if (j < 2 * m1) {
                const float f = (p.mask && !row_survives(j, m1, H)) ? 0.f : p.scale;
                out[corner_major_offset(j, c, m1, p.m2, p.m3)] = make_float2(Xr[mt][r] * f, Xi[mt][r] * f);
            }
        }
}

// K6: in (n_img, 4, m1, m2, m3) c64 -> out (n_img, H, C) c64
template <int JT>
__global__ __launch_bounds__(256) void cdft_inv_kernel(CdftParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* sTw = reinterpret_cast<float2*>(smem);
    const int H = p.H, C = p.C, m1 = p.m1;
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int n = tid; n < H; n += blockDim.x) sTw[n] = p.tw[n];
    __syncthreads();
    const int ctile = blockIdx.y * 4 + wave;
    if (ctile * 16 >= C) return;
    const int c = ctile * 16 + r16;
    const bool cvalid = c < C;
    const unsigned H8 = 8u * H;
    constexpr int KSJ = 4 * JT;
    const float2* in = reinterpret_cast<const float2*>(p.in) + (size_t)blockIdx.x * 4 * m1 * p.m2 * p.m3;

    // B operand, resident for the whole image: O[j = 4 ks + kk][c]
    float Or[KSJ], Oi[KSJ];
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) {
        const int j = 4 * ks + kk;
        float2 v = make_float2(0.f, 0.f);
        if (cvalid && j < 2 * m1 && !(p.mask && !row_survives(j, m1, H))) v = in[corner_major_offset(j, c, m1, p.m2, p.m3)];
        Or[ks] = v.x * p.scale;
        Oi[ks] = v.y * p.scale;
    }
    float2* out = reinterpret_cast<float2*>(p.out) + (size_t)blockIdx.x * H * C;
    const int nrt = (H + 15) >> 4;
    for (int rt = 0; rt < nrt; ++rt) {
        // A operand: e^{+i theta(j, h)}, lane: row h = 16 rt + r16, k-slot kk (j = 4 ks + kk)
        const int hA = min(16 * rt + r16, H - 1);
        const unsigned a4 = 8u * (unsigned)((4 * hA) % H);
        const unsigned b2 = 8u * (unsigned)(((long long)2 * m1 * hA) % H);
        unsigned aj = 8u * (unsigned)((kk * hA) % H);
        f32x4 Zr = f32x4{0, 0, 0, 0}, Zi = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSJ; ++ks) {
            const int j = 4 * ks + kk;
            unsigned id = (j >= m1) ? wrap_sub(aj, b2, H8) : aj;
            if (p.rowfreq) id = 8u * (unsigned)(((long long)p.rowfreq[min(j, 2 * m1 - 1)] * hA) % H);
            const float2 tw = lds_tw(sTw, id);
            // e^{+i theta} (or + i oi) = (c or - s oi) + i (c oi + s or)
            Zr = mfma16(tw.x, Or[ks], Zr);
            Zi = mfma16(tw.x, Oi[ks], Zi);
            Zr = mfma16(-tw.y, Oi[ks], Zr);
            Zi = mfma16(tw.y, Or[ks], Zi);
            aj = wrap_add(aj, a4, H8);
        }
        if (cvalid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = 16 * rt + 4 * kk + r;
                if (h < H) out[(size_t)h * C + c] = make_float2(Zr[r], Zi[r]);
            }
        }
    }
}

template <int T, bool INV>
static int launch_cdft_t(const CdftParams& p, hipStream_t s) {
    const size_t lds = (size_t)p.H * sizeof(float2);
    if (lds > 64 * 1024) { set_error("cdft: axis length %d too large", p.H); return -3; }
    const int ctiles = (p.C + 15) / 16;
    dim3 grid(p.n_img, (ctiles + 3) / 4);
    char name[64];
    const double bytes = (double)p.n_img * 8.0 * ((double)p.H * p.C + 4.0 * p.m1 * p.m2 * p.m3);
    if (INV) {
        snprintf(name, sizeof(name), "uno::cdft_inv_kernel<%d>", T);
        ProfScope prof(name, bytes, s);
        hipLaunchKernelGGL((cdft_inv_kernel<T>), grid, dim3(256), lds, s, p);
    } else {
        snprintf(name, sizeof(name), "uno::cdft_fwd_kernel<%d>", T);
        ProfScope prof(name, bytes, s);
        hipLaunchKernelGGL((cdft_fwd_kernel<T>), grid, dim3(256), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("cdft launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_cdft(const CdftParams& p, bool inverse, hipStream_t s) {
    const int T = (2 * p.m1 + 15) / 16;
#define UNO_CASE(t) if (T == t) return inverse ? launch_cdft_t<t, true>(p, s) : launch_cdft_t<t, false>(p, s);
    UNO_CASE(1) UNO_CASE(2) UNO_CASE(3) UNO_CASE(4) UNO_CASE(5)
#undef UNO_CASE
    set_error("cdft: modes1=%d exceeds the compiled range (<= 40)", p.m1);
    return -2;
}

}  // namespace uno
