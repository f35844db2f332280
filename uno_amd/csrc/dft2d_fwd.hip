// K1 - pruned forward 2-D DFT:  x (n_img, H, W) f32  ->  X (n_img, 2*m1, m2) c64
//
//   X[j][l] = scale * c_l * keep_j * sum_{h,w} x[h][w] * exp(-2 pi i (K_j h / H + l w / W))
//
// i.e. torch.fft.rfft2 restricted to the 2*m1 x m2 spectrum entries that
// SpectralConv2d_Uno.forward reads (reference integral_operators.py:187,198-203); the full
// half-spectrum is never materialised.  The same kernel computes gO = c (.) DFT_trunc(gy) in backward.
//
// Work decomposition: one workgroup per image, one wave per 16-row tile (tiles round-robin over waves).
//   stage A (rows, real -> m2 complex): symmetric form  E = x[w] + x[W-w], D = x[w] - x[W-w],
//     Tr = sum E cos, -Ti = sum D sin  -> half the flops of the plain real DFT.  Runs on
//     v_mfma_f32_16x16x4_f32 with M = 16 image rows, N = modes, K = column pairs.  The A operand comes
//     straight from global memory (each lane owns 8 consecutive columns of its row, k-order is free),
//     the B operand (twiddles) is gathered from a W-entry LDS table by (w*l mod W).
//   stage B (columns): X[j][l] += F[j][h] T[h][l] with M = corner rows, K = the tile's 16 rows; the
//     stage-A accumulator registers ARE the B operand (register r of lane-group g is row 4g+r), so the
//     intermediate never leaves registers.
// Each wave keeps a partial X for its tiles; a tree reduction through LDS (fixed order: deterministic)
// combines them and wave 0 writes the 2*m1*m2 complex results.
#include "uno_common.h"
#include <cstdio>

namespace uno {

constexpr int TAILMAX = 9;      // tail <= 31 pairs + w=0 + Nyquist column = 33 elements = 9 k-steps

template <int NT, int MT>
__global__ __launch_bounds__(256) void dft2d_fwd_kernel(Dft2dParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    float2* sTwW = reinterpret_cast<float2*>(smem);
    float2* sTwH = sTwW + W;
    unsigned* sTail = reinterpret_cast<unsigned*>(sTwH + H);        // [TAILMAX][NT][64] byte offsets
    float* sRed = reinterpret_cast<float*>(sTail + TAILMAX * NT * 64);

    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int NW = nthreads >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const unsigned W8 = 8u * W, H8 = 8u * H;

    // column-pair bookkeeping: pairs (w, W-w), w = 1..P; singles w = 0 and (W even) w = W/2
    const int P = (W - 1) >> 1;
    const int nfull = P >> 5;                   // chunks of 32 pairs handled by the vector path
    const int prem = P - (nfull << 5);
    const int ntail = prem + 1 + ((W & 1) ? 0 : 1);
    const int tailsteps = (ntail + 3) >> 2;

    for (int n = tid; n < W; n += nthreads) sTwW[n] = p.twW[n];
    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];
    for (int e = tid; e < TAILMAX * NT * 64; e += nthreads) {
        const int ln = e & 63, t = (e >> 6) % NT, s = e / (64 * NT);
        const int q = 4 * s + (ln >> 4);
        int w = 0;
        if (q < prem) w = 1 + 32 * nfull + q;
        else if (q == prem + 1 && !(W & 1)) w = W >> 1;
        const int l = min(16 * t + (ln & 15), m2 - 1);
        sTail[e] = 8u * (unsigned)(((long long)w * l) % W);
    }
    __syncthreads();

    // per-lane twiddle walk state for the vector path (B operand: k = kk, column = mode l)
    unsigned idx0[NT], stepL[NT], jump[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int l = min(16 * t + r16, m2 - 1);
        idx0[t] = 8u * (unsigned)(((1 + 8 * kk) * l) % W);
        stepL[t] = 8u * (unsigned)l;
        jump[t] = 8u * (unsigned)((24 * l) % W);
    }
    // stage-B A operand rows (corner rows) owned by this lane
    int Kj[MT];
    bool jvalid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int j = 16 * mt + r16;
        jvalid[mt] = j < 2 * m1;
        Kj[mt] = jvalid[mt] ? corner_freq(j, m1, H) : 0;
    }

    f32x4 Xr[MT][NT], Xi[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) { Xr[mt][t] = f32x4{0, 0, 0, 0}; Xi[mt][t] = f32x4{0, 0, 0, 0}; }

    const float* img = p.in + (size_t)blockIdx.x * H * W;
    const int nrt = (H + 15) >> 4;

    float L[8], R[8], Ln[8], Rn[8];
    auto row_ptr = [&](int rt) { return img + (size_t)min(rt * 16 + r16, H - 1) * W; };
    auto load_chunk = [&](float (&l)[8], float (&r)[8], const float* xr, int c) {
        const int a = 32 * c + 8 * kk;
        const f4u l0 = *reinterpret_cast<const f4u*>(xr + 1 + a);
        const f4u l1 = *reinterpret_cast<const f4u*>(xr + 5 + a);
        const f4u r0 = *reinterpret_cast<const f4u*>(xr + W - 8 - a);
        const f4u r1 = *reinterpret_cast<const f4u*>(xr + W - 4 - a);
#pragma unroll
        for (int s = 0; s < 4; ++s) { l[s] = l0.v[s]; l[4 + s] = l1.v[s]; r[s] = r0.v[s]; r[4 + s] = r1.v[s]; }
    };

    int rt = wave;
    if (rt < nrt && nfull > 0) load_chunk(L, R, row_ptr(rt), 0);

    for (; rt < nrt; rt += NW) {
        const float* xr = row_ptr(rt);
        // tail elements: issued early, consumed after the vector chunks
        float TL[TAILMAX], TR[TAILMAX];
#pragma unroll
        for (int s = 0; s < TAILMAX; ++s) {
            TL[s] = 0.f; TR[s] = 0.f;
            if (s < tailsteps) {
                const int q = 4 * s + kk;
                if (q < prem) {
                    const int w = 1 + 32 * nfull + q;
                    TL[s] = xr[w]; TR[s] = xr[W - w];
                } else if (q == prem) {
                    TL[s] = xr[0];
                } else if (q == prem + 1 && !(W & 1)) {
                    TL[s] = xr[W >> 1];
                }
            }
        }

        f32x4 Tr[NT], Tn[NT];           // Tn = -Im T
        unsigned idx[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { Tr[t] = f32x4{0, 0, 0, 0}; Tn[t] = f32x4{0, 0, 0, 0}; idx[t] = idx0[t]; }

        for (int c = 0; c < nfull; ++c) {
            if (c + 1 < nfull) load_chunk(Ln, Rn, xr, c + 1);
            else if (rt + NW < nrt) load_chunk(Ln, Rn, row_ptr(rt + NW), 0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float E = L[s] + R[7 - s];
                const float D = L[s] - R[7 - s];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float2 tw = lds_tw(sTwW, idx[t]);
                    Tr[t] = mfma16(E, tw.x, Tr[t]);
                    Tn[t] = mfma16(D, tw.y, Tn[t]);
                    idx[t] = wrap_add(idx[t], stepL[t], W8);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) idx[t] = wrap_add(idx[t], jump[t], W8);
#pragma unroll
            for (int s = 0; s < 8; ++s) { L[s] = Ln[s]; R[s] = Rn[s]; }
        }
#pragma unroll
        for (int s = 0; s < TAILMAX; ++s) {
            if (s < tailsteps) {
                const float E = TL[s] + TR[s];
                const float D = TL[s] - TR[s];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float2 tw = lds_tw(sTwW, sTail[(s * NT + t) * 64 + lane]);
                    Tr[t] = mfma16(E, tw.x, Tr[t]);
                    Tn[t] = mfma16(D, tw.y, Tn[t]);
                }
            }
        }

        // stage B: X[j][l] += exp(-i theta(j,h)) * T[h][l], h = 16 rt + 4 kk + s
        unsigned idxB[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            idxB[mt] = 8u * (unsigned)(((long long)Kj[mt] * (16 * rt + 4 * kk)) % H);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool hvalid = (16 * rt + 4 * kk + s) < H;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float2 tw = lds_tw(sTwH, idxB[mt]);
                const bool v = hvalid && jvalid[mt];
                const float ac = v ? tw.x : 0.f;
                const float ans = v ? -tw.y : 0.f;
                const float anc = -ac;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    Xr[mt][t] = mfma16(ac, Tr[t][s], Xr[mt][t]);
                    Xi[mt][t] = mfma16(anc, Tn[t][s], Xi[mt][t]);
                    Xr[mt][t] = mfma16(ans, Tn[t][s], Xr[mt][t]);
                    Xi[mt][t] = mfma16(ans, Tr[t][s], Xi[mt][t]);
                }
                idxB[mt] = wrap_add(idxB[mt], 8u * (unsigned)Kj[mt], H8);
            }
        }
    }

    // deterministic tree reduction of the per-wave partial spectra through LDS
    constexpr int NACC = MT * NT * 8;
    for (int stride = 2; stride >= 1; stride >>= 1) {
        if (stride >= NW) continue;
        if (wave >= stride && wave < 2 * stride) {
            float* slot = sRed + (size_t)(wave - stride) * NACC * 64;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        slot[((mt * NT + t) * 8 + r) * 64 + lane] = Xr[mt][t][r];
                        slot[((mt * NT + t) * 8 + 4 + r) * 64 + lane] = Xi[mt][t][r];
                    }
        }
        __syncthreads();
        if (wave < stride && wave + stride < NW) {
            const float* slot = sRed + (size_t)wave * NACC * 64;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        Xr[mt][t][r] += slot[((mt * NT + t) * 8 + r) * 64 + lane];
                        Xi[mt][t][r] += slot[((mt * NT + t) * 8 + 4 + r) * 64 + lane];
                    }
        }
        __syncthreads();
    }

    if (wave == 0) {
        float2* out = reinterpret_cast<float2*>(p.out) + (size_t)blockIdx.x * 2 * m1 * m2;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int l = 16 * t + r16;
            if (l >= m2) continue;
            const float cs = p.scale * (p.herm ? herm_weight(l, W) : 1.0f);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * mt + 4 * kk + r;
                    if (j < 2 * m1) {
                        const float f = (p.mask && !row_survives(j, m1, H)) ? 0.f : cs;
                        out[(size_t)j * m2 + l] = make_float2(Xr[mt][t][r] * f, Xi[mt][t][r] * f);
                    }
                }
        }
    }
}

template <int NT, int MT>
static int launch_fwd_t(const Dft2dParams& p, hipStream_t s) {
    const int nrt = (p.H + 15) / 16;
    const int NW = nrt >= 4 ? 4 : (nrt >= 2 ? 2 : 1);
    const size_t red = (size_t)(NW / 2) * MT * NT * 8 * 64 * sizeof(float);
    const size_t lds = (size_t)(p.W + p.H) * sizeof(float2) + (size_t)TAILMAX * NT * 64 * 4 + red;
    if (lds > 160 * 1024) { set_error("dft2d_fwd: grid %dx%d needs %zu B of LDS", p.H, p.W, lds); return -3; }
    auto k = dft2d_fwd_kernel<NT, MT>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("dft2d_fwd: cannot raise dynamic LDS to %zu", lds);
            return -4;
        }
    }
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_fwd_kernel<%d, %d>", NT, MT);
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * 4.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3(p.n_img), dim3(64 * NW), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_fwd launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_dft2d_fwd(const Dft2dParams& p, hipStream_t s) {
    const int NT = (p.m2 + 15) / 16, MT = (2 * p.m1 + 15) / 16;
#define UNO_CASE(nt, mt) if (NT == nt && MT == mt) return launch_fwd_t<nt, mt>(p, s);
    UNO_CASE(1, 1) UNO_CASE(1, 2) UNO_CASE(1, 3) UNO_CASE(1, 4) UNO_CASE(1, 5)
    UNO_CASE(2, 1) UNO_CASE(2, 2) UNO_CASE(2, 3) UNO_CASE(2, 4) UNO_CASE(2, 5)
    UNO_CASE(3, 1) UNO_CASE(3, 2) UNO_CASE(3, 3) UNO_CASE(3, 4) UNO_CASE(3, 5)
#undef UNO_CASE
    set_error("dft2d_fwd: modes (%d, %d) exceed the compiled range (modes1 <= 40, modes2 <= 48)", p.m1, p.m2);
    return -2;
}

}  // namespace uno
