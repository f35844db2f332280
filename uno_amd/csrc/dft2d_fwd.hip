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

constexpr int TAILMAX = 5;      // tail <= 15 pairs + w=0 + Nyquist column = 17 elements = 5 k-steps

// waves per SIMD the register allocator is asked to fit (accumulators: 8 NT MT for X + 8 NT for T)
template <int NT, int MT>
constexpr int fwd_waves_per_simd() {
    constexpr int acc = 8 * NT * MT + 8 * NT;
    return acc <= 16 ? 4 : (acc <= 48 ? 3 : (acc <= 104 ? 2 : 1));
}

template <int NT, int MT, bool VEC>
__global__ __launch_bounds__(256, (fwd_waves_per_simd<NT, MT>())) void dft2d_fwd_kernel(Dft2dParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    float2* sTwW = reinterpret_cast<float2*>(smem);
    float2* sTwH = sTwW + W;
    float* sRed = reinterpret_cast<float*>(sTwH + H);

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
    const int nfull = P >> 4;                   // chunks of 16 pairs handled by the vector path
    const int prem = P - (nfull << 4);
    const int ntail = prem + 1 + ((W & 1) ? 0 : 1);
    const int tailsteps = (ntail + 3) >> 2;

    for (int n = tid; n < W; n += nthreads) sTwW[n] = p.twW[n];
    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];
    __syncthreads();

    // per-lane twiddle walk of the vector path (B operand: k-slot kk, column = mode l): this lane owns
    // column pairs w = 1 + 16 c + 4 kk + s, s = 0..3, of chunk c
    unsigned idx0[NT], stepL[NT], jump[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int l = min(16 * t + r16, m2 - 1);
        idx0[t] = 8u * (unsigned)(((1 + 4 * kk) * l) % W);
        stepL[t] = 8u * (unsigned)l;
        jump[t] = 8u * (unsigned)((13 * l) % W);      // step from the last column pair of a chunk to the first of the next
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

    // Ring of four chunk buffers: chunk c of a row tile lives in buffer c & 3, loads run three chunks ahead.
    // Every load below is UNCONDITIONAL (chunk / row indices are clamped instead of branched around) so the
    // compiler can count outstanding loads and emit s_waitcnt vmcnt(N) with N > 0; a load under a branch
    // makes it fall back to vmcnt(0), which would expose the full HBM latency once per chunk.
    f4u bl[4], br[4];
    auto row_ptr = [&](int rt) { return img + (size_t)min(rt * 16 + r16, H - 1) * W; };
    const int clast = max(nfull - 1, 0);
#define UNO_LOAD_CHUNK(buf, xr, c)                                                        \
    do {                                                                                  \
        const int a_ = 16 * min((c), clast) + 4 * kk;                                     \
        bl[buf] = *reinterpret_cast<const f4u*>((xr) + 1 + a_);                           \
        br[buf] = *reinterpret_cast<const f4u*>((xr) + W - 4 - a_);                       \
        __builtin_amdgcn_sched_barrier(0);  /* keep the prefetch where it is issued */    \
    } while (0)

    // tail element of k-step s owned by this lane (-1 = none) and its twiddle index per mode tile
    int tlw[TAILMAX], trw[TAILMAX];
    unsigned tailidx[TAILMAX][NT];
#pragma unroll
    for (int s = 0; s < TAILMAX; ++s) {
        const int q = 4 * s + kk;
        const bool pair = q < prem;
        const bool nyq = (q == prem + 1) && !(W & 1);
        const int w = pair ? 1 + 16 * nfull + q : (nyq ? (W >> 1) : 0);
        tlw[s] = (pair || q == prem || nyq) ? w : -1;
        trw[s] = pair ? W - w : -1;
#pragma unroll
        for (int t = 0; t < NT; ++t) tailidx[s][t] = 8u * (unsigned)(((long long)w * (stepL[t] >> 3)) % W);
    }

    int rt = wave;
    if constexpr (VEC) {
        const float* xr0 = row_ptr(min(rt, nrt - 1));
        UNO_LOAD_CHUNK(0, xr0, 0);
        UNO_LOAD_CHUNK(1, xr0, 1);
        UNO_LOAD_CHUNK(2, xr0, 2);
    }

    for (; rt < nrt; rt += NW) {
        const float* xr = row_ptr(rt);
        float TL[TAILMAX], TR[TAILMAX];
#pragma unroll
        for (int s = 0; s < TAILMAX; ++s) {
            const float vl = xr[max(tlw[s], 0)];
            const float vr = xr[max(trw[s], 0)];
            TL[s] = tlw[s] >= 0 ? vl : 0.f;
            TR[s] = trw[s] >= 0 ? vr : 0.f;
        }

        f32x4 Tr[NT], Tn[NT];           // Tn = -Im T
#pragma unroll
        for (int t = 0; t < NT; ++t) { Tr[t] = f32x4{0, 0, 0, 0}; Tn[t] = f32x4{0, 0, 0, 0}; }

        if constexpr (VEC) {
            // software-pipelined twiddle gather: tw = twiddles of the step being multiplied, idx = table
            // offset of the step after it (LDS latency hides behind the current step's MFMAs)
            unsigned idx[NT];
            float2 tw[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                tw[t] = lds_tw(sTwW, idx0[t]);
                idx[t] = wrap_add(idx0[t], stepL[t], W8);
            }
#define UNO_COMPUTE_CHUNK(buf)                                                            \
    do {                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                   \
            const float E = bl[buf].v[s] + br[buf].v[3 - s];                              \
            const float D = bl[buf].v[s] - br[buf].v[3 - s];                              \
            float2 twn[NT];                                                               \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                              \
                twn[t] = lds_tw(sTwW, idx[t]);                                            \
                idx[t] = wrap_add(idx[t], s == 2 ? jump[t] : stepL[t], W8);               \
            }                                                                             \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                              \
                Tr[t] = mfma16(E, tw[t].x, Tr[t]);                                        \
                Tn[t] = mfma16(D, tw[t].y, Tn[t]);                                        \
            }                                                                             \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) tw[t] = twn[t];                \
        }                                                                                 \
    } while (0)

            int c = 0;
            for (; c + 4 <= nfull; c += 4) {
                UNO_LOAD_CHUNK(3, xr, c + 3);
                UNO_COMPUTE_CHUNK(0);
                UNO_LOAD_CHUNK(0, xr, c + 4);
                UNO_COMPUTE_CHUNK(1);
                UNO_LOAD_CHUNK(1, xr, c + 5);
                UNO_COMPUTE_CHUNK(2);
                UNO_LOAD_CHUNK(2, xr, c + 6);
                UNO_COMPUTE_CHUNK(3);
            }
            // 0..3 remaining chunks are already in buffers 0..2 (prefetched by the last group / the row prologue)
            const int rem = nfull - c;
            if (rem > 0) UNO_COMPUTE_CHUNK(0);
            if (rem > 1) UNO_COMPUTE_CHUNK(1);
            if (rem > 2) UNO_COMPUTE_CHUNK(2);
        }
        {
            float2 twt[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) twt[t] = lds_tw(sTwW, tailidx[0][t]);
#pragma unroll
            for (int s = 0; s < TAILMAX; ++s) {
                float2 twn[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) twn[t] = lds_tw(sTwW, tailidx[s + 1 < TAILMAX ? s + 1 : s][t]);
                if (s < tailsteps) {
                    const float E = TL[s] + TR[s];
                    const float D = TL[s] - TR[s];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        Tr[t] = mfma16(E, twt[t].x, Tr[t]);
                        Tn[t] = mfma16(D, twt[t].y, Tn[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) twt[t] = twn[t];
            }
        }
        if constexpr (VEC) {
            // all buffers are free: start the next row tile's first chunks, they land during stage B
            const float* xn = row_ptr(min(rt + NW, nrt - 1));
            UNO_LOAD_CHUNK(0, xn, 0);
            UNO_LOAD_CHUNK(1, xn, 1);
            UNO_LOAD_CHUNK(2, xn, 2);
        }

        // stage B: X[j][l] += exp(-i theta(j,h)) * T[h][l], h = 16 rt + 4 kk + s
        unsigned idxB[MT];
        float2 twB[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned i0 = 8u * (unsigned)(((long long)Kj[mt] * (16 * rt + 4 * kk)) % H);
            twB[mt] = lds_tw(sTwH, i0);
            idxB[mt] = wrap_add(i0, 8u * (unsigned)Kj[mt], H8);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool hvalid = (16 * rt + 4 * kk + s) < H;
            float2 twBn[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                twBn[mt] = lds_tw(sTwH, idxB[mt]);
                idxB[mt] = wrap_add(idxB[mt], 8u * (unsigned)Kj[mt], H8);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool v = hvalid && jvalid[mt];
                const float ac = v ? twB[mt].x : 0.f;
                const float ans = v ? -twB[mt].y : 0.f;
                const float anc = -ac;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    Xr[mt][t] = mfma16(ac, Tr[t][s], Xr[mt][t]);
                    Xi[mt][t] = mfma16(anc, Tn[t][s], Xi[mt][t]);
                    Xr[mt][t] = mfma16(ans, Tn[t][s], Xr[mt][t]);
                    Xi[mt][t] = mfma16(ans, Tr[t][s], Xi[mt][t]);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) twB[mt] = twBn[mt];
        }
    }
#undef UNO_LOAD_CHUNK
#undef UNO_COMPUTE_CHUNK

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

template <int NT, int MT, bool VEC>
static int launch_fwd_t(const Dft2dParams& p, hipStream_t s) {
    const int nrt = (p.H + 15) / 16;
    const int NW = nrt >= 4 ? 4 : (nrt >= 2 ? 2 : 1);
    const size_t red = (size_t)(NW / 2) * MT * NT * 8 * 64 * sizeof(float);
    const size_t lds = (size_t)(p.W + p.H) * sizeof(float2) + red;
    if (lds > 160 * 1024) { set_error("dft2d_fwd: grid %dx%d needs %zu B of LDS", p.H, p.W, lds); return -3; }
    auto k = dft2d_fwd_kernel<NT, MT, VEC>;
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
    const bool vec = ((p.W - 1) >> 1) >= 16;       // at least one full chunk of 16 column pairs
#define UNO_CASE(nt, mt) if (NT == nt && MT == mt) return vec ? launch_fwd_t<nt, mt, true>(p, s) : launch_fwd_t<nt, mt, false>(p, s);
    UNO_CASE(1, 1) UNO_CASE(1, 2) UNO_CASE(1, 3) UNO_CASE(1, 4) UNO_CASE(1, 5)
    UNO_CASE(2, 1) UNO_CASE(2, 2) UNO_CASE(2, 3) UNO_CASE(2, 4) UNO_CASE(2, 5)
    UNO_CASE(3, 1) UNO_CASE(3, 2) UNO_CASE(3, 3) UNO_CASE(3, 4) UNO_CASE(3, 5)
#undef UNO_CASE
    set_error("dft2d_fwd: modes (%d, %d) exceed the compiled range (modes1 <= 40, modes2 <= 48)", p.m1, p.m2);
    return -2;
}

}  // namespace uno
