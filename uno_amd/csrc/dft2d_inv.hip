// K3 - pruned inverse 2-D DFT:  O (n_img, 2*m1, m2) c64  ->  y (n_img, H, W) f32
//
//   y[h][w] = Re sum_{j,l} scale * c_l * keep_j * O[j][l] * exp(+2 pi i (K_j h / H + l w / W))
//
// i.e. torch.fft.irfft2(out_ft, s=(H, W), norm="forward") of a spectrum that is zero outside the two
// low-frequency corners (reference integral_operators.py:190-206) - the zero-filled out_ft is never
// materialised.  With herm=0, mask=0 and scale=1/(H W) it is the gx stage of the backward pass.
//
// One workgroup per image, one wave per 16-row tile of the output.
//   stage B' (columns): U^T[l][h] = sum_j O[l][j] exp(+i theta(j,h)):  M = modes, N = the tile's 16
//     rows, K = corner rows.  The O operand is loaded once per image into registers (A operand); the
//     mode <-> M-row assignment is permuted (row 4g+r computes mode 4r+g) so that ...
//   stage A' (rows): ... the stage-B' accumulators are directly the A operand of the row transform
//     (k-step s, lane group kk <-> mode 4s+kk).  Symmetric form: Ey = sum Ur cos, Dy = sum Ui sin over
//     w <= W/2, then y[w] = Ey - Dy and y[W-w] = Ey + Dy: half the flops, no padding waste in K.
#include "uno_common.h"
#include <cstdio>

namespace uno {

template <int NT, int JT>
__global__ __launch_bounds__(256) void dft2d_inv_kernel(Dft2dParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    float2* sTwW = reinterpret_cast<float2*>(smem);
    float2* sTwH = sTwW + W;

    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int NW = nthreads >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const unsigned W8 = 8u * W, H8 = 8u * H;
    constexpr int KSJ = 4 * JT;                 // k-steps over corner rows
    constexpr int KSA = 4 * NT;                 // upper bound of k-steps over modes
    const int ksa = (m2 + 3) >> 2;              // k-steps actually needed

    for (int n = tid; n < W; n += nthreads) sTwW[n] = p.twW[n];
    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];

    // stage-B' A operand: O[mode(rho)][j = 4 ks + kk], rho = r16, mode = 16 t + 4 (rho & 3) + (rho >> 2)
    const float2* O = reinterpret_cast<const float2*>(p.in) + (size_t)blockIdx.x * 2 * m1 * m2;
    float Or[NT][KSJ], Oi[NT][KSJ];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int l = 16 * t + 4 * (r16 & 3) + (r16 >> 2);
        const float cs = p.scale * ((p.herm && l < m2) ? herm_weight(l, W) : 1.0f);
#pragma unroll
        for (int ks = 0; ks < KSJ; ++ks) {
            const int j = 4 * ks + kk;
            float2 v = make_float2(0.f, 0.f);
            if (l < m2 && j < 2 * m1 && !(p.mask && !row_survives(j, m1, H))) v = O[(size_t)j * m2 + l];
            Or[t][ks] = v.x * cs;
            Oi[t][ks] = v.y * cs;
        }
    }
    // stage-A' twiddle walk: k-step sp covers mode l = 4 sp + kk at column w = 16 wt + r16
    unsigned idxA0[KSA], stepA[KSA];
#pragma unroll
    for (int sp = 0; sp < KSA; ++sp) {
        const int l = min(4 * sp + kk, m2 - 1);
        idxA0[sp] = 8u * (unsigned)((l * r16) % W);
        stepA[sp] = 8u * (unsigned)((16 * l) % W);
    }
    __syncthreads();

    float* img = p.out + (size_t)blockIdx.x * H * W;
    const int nrt = (H + 15) >> 4;
    const int NE = (W >> 1) + 1;                // columns 0..W/2 are computed, the rest mirrored
    const int nwt = (NE + 15) >> 4;

    for (int rt = wave; rt < nrt; rt += NW) {
        // ---- stage B': B operand = exp(+i theta), theta = 2 pi K_j h / H, lane: k = kk (j = 4ks+kk), col = h
        const int hB = min(16 * rt + r16, H - 1);
        const unsigned a4 = 8u * (unsigned)((4 * hB) % H);               // advance of (j h mod H) per k-step
        const unsigned b2 = 8u * (unsigned)(((long long)2 * m1 * hB) % H);   // (2 m1 h) mod H
        unsigned aj = 8u * (unsigned)((kk * hB) % H);                    // (j h) mod H, j = kk
        f32x4 Ur[NT], Ui[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { Ur[t] = f32x4{0, 0, 0, 0}; Ui[t] = f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int ks = 0; ks < KSJ; ++ks) {
            const int j = 4 * ks + kk;
            // K_j = j for the lo corner, j - 2 m1 (mod H) for the hi corner
            const unsigned id = (j >= m1) ? wrap_sub(aj, b2, H8) : aj;
            const float2 tw = lds_tw(sTwH, id);
            const float ns = -tw.y;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                Ur[t] = mfma16(Or[t][ks], tw.x, Ur[t]);
                Ui[t] = mfma16(Or[t][ks], tw.y, Ui[t]);
                Ur[t] = mfma16(Oi[t][ks], ns, Ur[t]);
                Ui[t] = mfma16(Oi[t][ks], tw.x, Ui[t]);
            }
            aj = wrap_add(aj, a4, H8);
        }

        // ---- stage A': per 16-column tile, K = modes
        unsigned idxA[KSA];
#pragma unroll
        for (int sp = 0; sp < KSA; ++sp) idxA[sp] = idxA0[sp];
        for (int wt = 0; wt < nwt; ++wt) {
            f32x4 Ey = f32x4{0, 0, 0, 0}, Dy = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int sp = 0; sp < KSA; ++sp) {
                if (sp < ksa) {
                    const float2 tw = lds_tw(sTwW, idxA[sp]);
                    Ey = mfma16(Ur[sp >> 2][sp & 3], tw.x, Ey);
                    Dy = mfma16(Ui[sp >> 2][sp & 3], tw.y, Dy);
                    idxA[sp] = wrap_add(idxA[sp], stepA[sp], W8);
                }
            }
            const int w = 16 * wt + r16;
            const bool lv = w < NE;
            const bool rv = lv && w >= 1 && 2 * w != W;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = 16 * rt + 4 * kk + r;
                if (h < H) {
                    float* row = img + (size_t)h * W;
                    if (lv) row[w] = Ey[r] - Dy[r];
                    if (rv) row[W - w] = Ey[r] + Dy[r];
                }
            }
        }
    }
}

template <int NT, int JT>
static int launch_inv_t(const Dft2dParams& p, hipStream_t s) {
    const int nrt = (p.H + 15) / 16;
    const int NW = nrt >= 4 ? 4 : (nrt >= 2 ? 2 : 1);
    const size_t lds = (size_t)(p.W + p.H) * sizeof(float2);
    if (lds > 160 * 1024) { set_error("dft2d_inv: grid %dx%d needs %zu B of LDS", p.H, p.W, lds); return -3; }
    auto k = dft2d_inv_kernel<NT, JT>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("dft2d_inv: cannot raise dynamic LDS to %zu", lds);
            return -4;
        }
    }
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_inv_kernel<%d, %d>", NT, JT);
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * 4.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3(p.n_img), dim3(64 * NW), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_inv launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_dft2d_inv(const Dft2dParams& p, hipStream_t s) {
    const int NT = (p.m2 + 15) / 16, JT = (2 * p.m1 + 15) / 16;
#define UNO_CASE(nt, jt) if (NT == nt && JT == jt) return launch_inv_t<nt, jt>(p, s);
    UNO_CASE(1, 1) UNO_CASE(1, 2) UNO_CASE(1, 3) UNO_CASE(1, 4) UNO_CASE(1, 5)
    UNO_CASE(2, 1) UNO_CASE(2, 2) UNO_CASE(2, 3) UNO_CASE(2, 4) UNO_CASE(2, 5)
    UNO_CASE(3, 1) UNO_CASE(3, 2) UNO_CASE(3, 3) UNO_CASE(3, 4) UNO_CASE(3, 5)
#undef UNO_CASE
    set_error("dft2d_inv: modes (%d, %d) exceed the compiled range (modes1 <= 40, modes2 <= 48)", p.m1, p.m2);
    return -2;
}

}  // namespace uno
