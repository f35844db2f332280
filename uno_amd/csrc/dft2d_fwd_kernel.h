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
#pragma once
#include "uno_common.h"
#include "dft2d_fwd_ft_kernel.h"
#include "dft2d_fwd_ht_kernel.h"
#include <cstdio>

namespace uno {

constexpr int TAILMAX = 5;      // tail <= 15 pairs + w=0 + Nyquist column = 17 elements = 5 k-steps

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    // v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products, block = lane / 4 (probed on gfx950:
    // tools/probes/mfma4x4_probe.hip): A[i] = lane 4*block + i, B[j] = lane 4*block + j, D[i][j] = lane 4*block + j, reg i
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// waves per SIMD the register allocator is asked to fit (accumulators: 8 NT MT for X + 8 NT for T)
template <int NT, int MT>
constexpr int fwd_waves_per_simd() {
    constexpr int acc = 8 * NT * MT + 8 * NT;
    return acc <= 16 ? 4 : (acc <= 64 ? 3 : (acc <= 104 ? 2 : 1));
}

// R4 > 0: the last mode tile holds at most 4*R4 <= 8 modes and is computed with R4 4x4x1 MFMAs (8 cycles each)
// instead of one 16x16x4 MFMA (32 cycles) that would be 50-94 % padding; "stream" q < NTF is a full 16-mode
// tile, stream NTF + g is the 4-mode group g.
// BF16: the images are bfloat16 (config C5: bf16 activations, f32 accumulation): 8-byte loads of four values, widened
// (<< 16) where they enter the E / D sums; the spectrum stays complex64.
template <int NT, int MT, bool VEC, int R4, bool BF16>
__global__ __launch_bounds__(256, (fwd_waves_per_simd<NT, MT>())) void dft2d_fwd_kernel(Dft2dParams p) {
    using in_t = typename IoElem<BF16>::type;           // float | unsigned short
    using vec_t = typename IoElem<BF16>::vec4;          // f4u | h4u: four consecutive elements, element-aligned
    constexpr int NTF = R4 > 0 ? NT - 1 : NT;
    constexpr int NS = NTF + R4;
    constexpr int NQ = R4 > 0 ? R4 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    float2* sTwW = reinterpret_cast<float2*>(smem);
    float2* sTwH = sTwW + W;
    int* sTailW = reinterpret_cast<int*>(sTwH + H);                  // [TAILMAX][2][64]: left / right column of a tail element (-1 = none)
    unsigned* sTailI = reinterpret_cast<unsigned*>(sTailW + TAILMAX * 2 * 64);   // [TAILMAX][NS][64]: its twiddle offset per stage-A stream
    float* sRed = reinterpret_cast<float*>(sTailI + TAILMAX * NS * 64);

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
    // tail element of k-step s, k-slot (lane >> 4): pairs beyond the last full chunk, then w = 0, then the Nyquist column
    for (int e = tid; e < TAILMAX * 64; e += nthreads) {
        const int ln = e & 63, sq = e >> 6;
        const int q = 4 * sq + (ln >> 4);
        const bool pair = q < prem;
        const bool nyq = (q == prem + 1) && !(W & 1);
        const int w = pair ? 1 + 16 * nfull + q : (nyq ? (W >> 1) : 0);
        sTailW[(sq * 2 + 0) * 64 + ln] = (pair || q == prem || nyq) ? w : -1;
        sTailW[(sq * 2 + 1) * 64 + ln] = pair ? W - w : -1;
        for (int t = 0; t < NS; ++t) {
            const int lm = t < NTF ? 16 * t + (ln & 15) : 16 * NTF + 4 * (t - NTF) + (ln & 3);
            const unsigned l = (unsigned)min(lm, m2 - 1);
            sTailI[(sq * NS + t) * 64 + ln] = 8u * (((unsigned)w * l) % (unsigned)W);
        }
    }
    __syncthreads();

    // per-lane twiddle walk of the vector path (B operand: k-slot kk, column = mode l): this lane owns
    // column pairs w = 1 + 16 c + 4 kk + s, s = 0..3, of chunk c
    unsigned idx0[NS], stepL[NS], jump[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int lm = t < NTF ? 16 * t + r16 : 16 * NTF + 4 * (t - NTF) + (lane & 3);
        const unsigned l = (unsigned)min(lm, m2 - 1);
        idx0[t] = 8u * (((1u + 4u * kk) * l) % (unsigned)W);
        stepL[t] = 8u * l;
        jump[t] = 8u * ((13u * l) % (unsigned)W);      // step from the last column pair of a chunk to the first of the next
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

    const in_t* img = reinterpret_cast<const in_t*>(p.in) + (size_t)blockIdx.x * H * W;
    const int nrt = (H + 15) >> 4;

    // Ring of four chunk buffers: chunk c of a row tile lives in buffer c & 3, loads run three chunks ahead.
    // Every load below is UNCONDITIONAL (chunk / row indices are clamped instead of branched around) so the
    // compiler can count outstanding loads and emit s_waitcnt vmcnt(N) with N > 0; a load under a branch
    // makes it fall back to vmcnt(0), which would expose the full HBM latency once per chunk.
    vec_t bl[4], br[4];
    auto row_ptr = [&](int rt) { return img + (size_t)min(rt * 16 + r16, H - 1) * W; };
    const int clast = max(nfull - 1, 0);
#define UNO_LOAD_CHUNK(buf, xr, c)                                                        \
    do {                                                                                  \
        const int a_ = 16 * min((c), clast) + 4 * kk;                                     \
        bl[buf] = *reinterpret_cast<const vec_t*>((xr) + 1 + a_);                         \
        br[buf] = *reinterpret_cast<const vec_t*>((xr) + W - 4 - a_);                     \
        __builtin_amdgcn_sched_barrier(0);  /* keep the prefetch where it is issued */    \
    } while (0)

    int rt = wave;
    if constexpr (VEC) {
        const in_t* xr0 = row_ptr(min(rt, nrt - 1));
        UNO_LOAD_CHUNK(0, xr0, 0);
        UNO_LOAD_CHUNK(1, xr0, 1);
        UNO_LOAD_CHUNK(2, xr0, 2);
    }

    for (; rt < nrt; rt += NW) {
        const in_t* xr = row_ptr(rt);
        asm volatile("" ::: "memory");          // keep the (loop-invariant) LDS table reads inside the loop: registers are scarcer
        float TL[TAILMAX], TR[TAILMAX];
#pragma unroll
        for (int s = 0; s < TAILMAX; ++s) {
            const int wl = sTailW[(s * 2 + 0) * 64 + lane], wr = sTailW[(s * 2 + 1) * 64 + lane];
            const float vl = io_widen(xr[max(wl, 0)]);    // unconditional (clamped) loads, masked by select
            const float vr = io_widen(xr[max(wr, 0)]);
            TL[s] = wl >= 0 ? vl : 0.f;
            TR[s] = wr >= 0 ? vr : 0.f;
        }

        f32x4 Tr[NT], Tn[NT];           // Tn = -Im T
        f32x4 Qr[NQ], Qn[NQ];           // 4x4x1 accumulators of the 4-mode groups (R4 > 0)
#pragma unroll
        for (int t = 0; t < NT; ++t) { Tr[t] = f32x4{0, 0, 0, 0}; Tn[t] = f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int g = 0; g < NQ; ++g) { Qr[g] = f32x4{0, 0, 0, 0}; Qn[g] = f32x4{0, 0, 0, 0}; }
#define UNO_STAGE_A_MFMA(E_, D_, TW_)                                                     \
    do {                                                                                  \
        _Pragma("unroll") for (int t = 0; t < NTF; ++t) {                                 \
            Tr[t] = mfma16((E_), (TW_)[t].x, Tr[t]);                                      \
            Tn[t] = mfma16((D_), (TW_)[t].y, Tn[t]);                                      \
        }                                                                                 \
        _Pragma("unroll") for (int g = 0; g < R4; ++g) {                                  \
            Qr[g] = mfma4((E_), (TW_)[NTF + g].x, Qr[g]);                                 \
            Qn[g] = mfma4((D_), (TW_)[NTF + g].y, Qn[g]);                                 \
        }                                                                                 \
    } while (0)

        if constexpr (VEC) {
            // software-pipelined twiddle gather: tw = twiddles of the step being multiplied, idx = table
            // offset of the step after it (LDS latency hides behind the current step's MFMAs)
            unsigned idx[NS];
            float2 tw[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                tw[t] = lds_tw(sTwW, idx0[t]);
                idx[t] = wrap_add(idx0[t], stepL[t], W8);
            }
#define UNO_COMPUTE_CHUNK(buf)                                                            \
    do {                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                   \
            const float E = io_widen(bl[buf].v[s]) + io_widen(br[buf].v[3 - s]);          \
            const float D = io_widen(bl[buf].v[s]) - io_widen(br[buf].v[3 - s]);          \
            float2 twn[NS];                                                               \
            _Pragma("unroll") for (int t = 0; t < NS; ++t) {                              \
                twn[t] = lds_tw(sTwW, idx[t]);                                            \
                idx[t] = wrap_add(idx[t], s == 2 ? jump[t] : stepL[t], W8);               \
            }                                                                             \
            UNO_STAGE_A_MFMA(E, D, tw);                                                   \
            _Pragma("unroll") for (int t = 0; t < NS; ++t) tw[t] = twn[t];                \
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
            float2 twt[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) twt[t] = lds_tw(sTwW, sTailI[t * 64 + lane]);
#pragma unroll
            for (int s = 0; s < TAILMAX; ++s) {
                float2 twn[NS];
#pragma unroll
                for (int t = 0; t < NS; ++t) twn[t] = lds_tw(sTwW, sTailI[((s + 1 < TAILMAX ? s + 1 : s) * NS + t) * 64 + lane]);
                if (s < tailsteps) {
                    const float E = TL[s] + TR[s];
                    const float D = TL[s] - TR[s];
                    UNO_STAGE_A_MFMA(E, D, twt);
                }
#pragma unroll
                for (int t = 0; t < NS; ++t) twt[t] = twn[t];
            }
        }
        if constexpr (VEC) {
            // all buffers are free: start the next row tile's first chunks, they land during stage B
            const in_t* xn = row_ptr(min(rt + NW, nrt - 1));
            UNO_LOAD_CHUNK(0, xn, 0);
            UNO_LOAD_CHUNK(1, xn, 1);
            UNO_LOAD_CHUNK(2, xn, 2);
        }

        if constexpr (R4 > 0) {
            // 4x4x1 result: lane 16 ws + 4 rg + j, reg i = partial T[row 4 rg + i][mode 16 NTF + 4 g + j] of k-slot ws.
            // Sum over the four k-slots, then move to the 16x16x4 accumulator layout stage B consumes
            // (lane (kk, n), reg s = T[row 4 kk + s][mode n]); columns n >= 4 R4 of the last tile are zero.
            f32x4 lastR = f32x4{0, 0, 0, 0}, lastN = f32x4{0, 0, 0, 0};
            const int src = 20 * kk + (r16 & 3);
#pragma unroll
            for (int g = 0; g < R4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float vr = Qr[g][i], vn = Qn[g][i];
                    vr += __shfl_xor(vr, 16); vn += __shfl_xor(vn, 16);
                    vr += __shfl_xor(vr, 32); vn += __shfl_xor(vn, 32);
                    const float gr = __shfl(vr, src), gn = __shfl(vn, src);
                    if ((r16 >> 2) == g) { lastR[i] = gr; lastN[i] = gn; }
                }
            Tr[NT - 1] = lastR;
            Tn[NT - 1] = lastN;
        }

        // stage B: X[j][l] += exp(-i theta(j,h)) * T[h][l], h = 16 rt + 4 kk + s
        unsigned idxB[MT];
        float2 twB[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned i0 = 8u * (((unsigned)Kj[mt] * (unsigned)(16 * rt + 4 * kk)) % (unsigned)H);
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
#undef UNO_STAGE_A_MFMA

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
        float2* out = reinterpret_cast<float2*>(p.out) + spectrum_index(p, blockIdx.x) * 2 * m1 * m2;
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

template <int NT, int MT, bool VEC, int R4, bool BF16>
static int launch_fwd_b(const Dft2dParams& p, hipStream_t s) {
    constexpr int NS = (R4 > 0 ? NT - 1 : NT) + R4;
    const int nrt = (p.H + 15) / 16;
    const int NW = (long long)p.H * p.W < 4096 ? 1 : pick_waves_per_image(nrt);     // small images (3-D planes): one wave each, more images in flight per CU
    const size_t red = (size_t)(NW / 2) * MT * NT * 8 * 64 * sizeof(float);
    const size_t lds = (size_t)(p.W + p.H) * sizeof(float2) + (size_t)TAILMAX * (2 + NS) * 64 * 4 + red;
    if (lds > 160 * 1024) { set_error("dft2d_fwd: grid %dx%d needs %zu B of LDS", p.H, p.W, lds); return -3; }
    auto k = dft2d_fwd_kernel<NT, MT, VEC, R4, BF16>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_slot)) { set_error("dft2d_fwd: cannot raise dynamic LDS to %zu", lds); return -4; }
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_fwd_kernel<%d, %d, %s, %d, %s>", NT, MT, VEC ? "true" : "false", R4, BF16 ? "true" : "false");
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * (BF16 ? 2.0 : 4.0) + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3(p.n_img), dim3(64 * NW), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_fwd launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

template <int NT, int MT, bool VEC, int R4>
static int launch_fwd_t(const Dft2dParams& p, hipStream_t s) {
    if constexpr (VEC) {
        FwdFtGeometry ft;
        if (fwd_ft_geometry(p, NT, MT, R4, &ft)) return launch_fwd_ft<NT, MT, R4>(p, ft, s);
        if (fwd_ht_geometry(p, NT, MT, R4, &ft)) return launch_fwd_ht<NT, MT, R4>(p, ft, s);
    }
    return p.bf16 ? launch_fwd_b<NT, MT, VEC, R4, true>(p, s) : launch_fwd_b<NT, MT, VEC, R4, false>(p, s);
}

}  // namespace uno
