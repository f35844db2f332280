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
//   stage A' (rows): ... the stage-B' accumulators are directly the B operand of the row transform
//     (k-step s, lane group kk <-> mode 4s+kk), with the twiddles as A operand: D[w][h], i.e. every lane
//     ends up with FOUR CONSECUTIVE COLUMNS of one row.  Symmetric form: Ey = sum Ur cos, Dy = sum Ui sin
//     over w <= W/2, then y[w] = Ey - Dy and y[W-w] = Ey + Dy: half the flops, no padding waste in K.
//   stores: a scattered 16-rows-per-instruction store pattern costs more than all the MFMAs (measured:
//     117 of 295 us), so each wave stages 64-column chunks of its 16 rows (left half and mirrored half)
//     in a private LDS buffer (ds_write_b128) and writes them out as 256-byte row segments with
//     16-byte-per-lane stores.
#include "uno_common.h"
#include <cstdio>

#ifndef UNO_ABLATE
#define UNO_ABLATE 0        // developer ablation builds only (tools/ablate.sh); 0 = product
#endif

namespace uno {

constexpr int STG_COLS = 64;            // columns per staged chunk (4 MFMA column tiles)
constexpr int STG_RS = STG_COLS + 4;    // LDS row stride in floats: 16-byte aligned, 4-bank skew per row

// KS = ceil(modes2 / 4): k-steps of the row stage (compile-time, so the MFMA chains are straight-line code);
// JT = ceil(2 modes1 / 16): 16-row tiles of corner rows.
template <int KS, int JT>
constexpr int inv_waves_per_simd() {
    constexpr int NT = (KS + 3) / 4;
    // resident P/M operand (4 NT (2 JT + 1)) + U accumulators (8 NT) + twiddle walk state (7 KS) + addressing etc.
    constexpr int regs = 4 * NT * (2 * JT + 1) + 8 * NT + 7 * KS + 44;
    return regs <= 120 ? 4 : (regs <= 160 ? 3 : (regs <= 230 ? 2 : 1));
}

// BF16: the images are written as bfloat16 (round to nearest even; config C5), everything before the store is f32.
template <int KS, int JT, bool BF16>
__global__ __launch_bounds__(256, (inv_waves_per_simd<KS, JT>())) void dft2d_inv_kernel(Dft2dParams p) {
    using out_t = typename IoElem<BF16>::type;
    constexpr int NT = (KS + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    float* sStage = reinterpret_cast<float*>(smem);                       // [NW][2][16][STG_RS]
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int NW = nthreads >> 6;
    float2* sTwW = reinterpret_cast<float2*>(sStage + NW * 2 * 16 * STG_RS);
    float2* sTwH = sTwW + W;
    unsigned* sIdxA0 = reinterpret_cast<unsigned*>(sTwH + H);          // [KS][64]: start of the stage-A' twiddle walk

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const unsigned W8 = 8u * W, H8 = 8u * H;
    constexpr int KSA = KS;                     // k-steps over modes

    for (int n = tid; n < W; n += nthreads) sTwW[n] = p.twW[n];
    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];

    // stage-B' A operand.  The corner rows come in +-k pairs (lo corner row k <-> frequency +k, hi corner row
    // 2 m1 - k <-> frequency -k), so with P_k = O[+k] + O[-k], M_k = O[+k] - O[-k]:
    //     U[h] = sum_{k=0}^{m1} cos(theta_k h) P_k + i sin(theta_k h) M_k          (theta_k = 2 pi k / H)
    // i.e. m1 + 1 real twiddle pairs instead of 2 m1 complex ones - 40 % fewer MFMAs in this stage.
    // Operand lane (rho = r16 -> mode 16 t + 4 (rho & 3) + (rho >> 2), k-slot kk -> k = 4 ks + kk).
    constexpr int KSK = 2 * JT + 1;                       // >= ceil((m1 + 1) / 4)
    const int ksk = (m1 + 4) >> 2;                        // k-steps actually needed
    const float2* O = reinterpret_cast<const float2*>(p.in) + spectrum_index(p, blockIdx.x) * 2 * m1 * m2;
    float Pr[NT][KSK], Pi[NT][KSK], Mr[NT][KSK], Mi[NT][KSK];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int l = 16 * t + 4 * (r16 & 3) + (r16 >> 2);
        const float cs = p.scale * ((p.herm && l < m2) ? herm_weight(l, W) : 1.0f);
#pragma unroll
        for (int ks = 0; ks < KSK; ++ks) {
            const int k = 4 * ks + kk;
            float2 vp = make_float2(0.f, 0.f), vm = make_float2(0.f, 0.f);
            if (l < m2 && k < m1 && !(p.mask && !row_survives(k, m1, H))) vp = O[(size_t)k * m2 + l];                 // +k: lo corner row k
            if (l < m2 && k >= 1 && k <= m1) vm = O[(size_t)(2 * m1 - k) * m2 + l];                                      // -k: hi corner row 2 m1 - k
            Pr[t][ks] = (vp.x + vm.x) * cs; Pi[t][ks] = (vp.y + vm.y) * cs;
            Mr[t][ks] = (vp.x - vm.x) * cs; Mi[t][ks] = (vp.y - vm.y) * cs;
        }
    }
    // stage-A' twiddle walk (A operand): lane (i = r16 -> column w = 16 wt + r16, k-slot kk -> mode 4 sp + kk)
    unsigned stepA[KSA];
#pragma unroll
    for (int sp = 0; sp < KSA; ++sp) {
        const unsigned l = (unsigned)min(4 * sp + kk, m2 - 1);
        stepA[sp] = 8u * ((16u * l) % (unsigned)W);
    }
    for (int e = tid; e < KSA * 64; e += nthreads) {
        const unsigned l = (unsigned)min(4 * (e >> 6) + ((e & 63) >> 4), m2 - 1);
        sIdxA0[e] = 8u * ((l * (unsigned)(e & 15)) % (unsigned)W);
    }
    __syncthreads();

    out_t* img = reinterpret_cast<out_t*>(p.out) + (size_t)blockIdx.x * H * W;
    const int nrt = (H + 15) >> 4;
    const int Wh = W >> 1;                      // columns 0..Wh are computed, Wh+1..W-1 are their mirror images
    const int nwt = (Wh + 16) >> 4;             // 16-column tiles covering 0..Wh
    const int nchunk = (nwt + 3) >> 2;
    float* stL = sStage + (size_t)wave * 2 * 16 * STG_RS;     // this wave's left-half chunk  [16][STG_RS]
    float* stR = stL + 16 * STG_RS;                            // mirrored-half chunk

    for (int rt = wave; rt < nrt; rt += NW) {
        // ---- stage B': B operand = (cos, sin)(2 pi k h / H), lane: k-slot kk (k = 4 ks + kk), column = row h of the tile
        const unsigned hB = (unsigned)min(16 * rt + r16, H - 1);
        const unsigned a4 = 8u * ((4u * hB) % (unsigned)H);               // advance of (k h mod H) per k-step
        unsigned aj = 8u * (((unsigned)kk * hB) % (unsigned)H);          // (k h) mod H, k = kk
        f32x4 Ur[NT], Ui[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { Ur[t] = f32x4{0, 0, 0, 0}; Ui[t] = f32x4{0, 0, 0, 0}; }
        float2 twb = lds_tw(sTwH, aj);
#pragma unroll
        for (int ks = 0; ks < KSK; ++ks) {
            aj = wrap_add(aj, a4, H8);
            const float2 twn = lds_tw(sTwH, aj);        // next k-step's twiddle (LDS latency hides behind the MFMAs)
            if (ks < ksk) {
                const float ns = -twb.y;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    Ur[t] = mfma16(Pr[t][ks], twb.x, Ur[t]);
                    Ui[t] = mfma16(Pi[t][ks], twb.x, Ui[t]);
                    Ur[t] = mfma16(Mi[t][ks], ns, Ur[t]);
                    Ui[t] = mfma16(Mr[t][ks], twb.y, Ui[t]);
                }
            }
            twb = twn;
        }

        // ---- stage A': D[w][h] = sum_modes tw[w][mode] * U[mode][h]; lane (h = r16, g = kk) gets columns 4g..4g+3
        unsigned idxA[KSA];
        float2 twa[KSA];
        asm volatile("" ::: "memory");          // keep the loop-invariant table read below inside the loop (register pressure)
#pragma unroll
        for (int sp = 0; sp < KSA; ++sp) {
            const unsigned i0 = sIdxA0[sp * 64 + lane];
            twa[sp] = lds_tw(sTwW, i0);
            idxA[sp] = wrap_add(i0, stepA[sp], W8);
        }
        for (int ch = 0; ch < nchunk; ++ch) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int wt = 4 * ch + t4;
                if (wt < nwt) {
                    float2 twn[KSA];
#pragma unroll
                    for (int sp = 0; sp < KSA; ++sp) {
                        if (true) {
                            twn[sp] = (UNO_ABLATE & 1) ? twa[sp] : lds_tw(sTwW, idxA[sp]);
                            idxA[sp] = wrap_add(idxA[sp], stepA[sp], W8);
                        }
                    }
                    f32x4 Ey = f32x4{0, 0, 0, 0}, Dy = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int sp = 0; sp < KSA; ++sp) {
                        if (true) {
                            Ey = mfma16(twa[sp].x, Ur[sp >> 2][sp & 3], Ey);
                            Dy = mfma16(twa[sp].y, Ui[sp >> 2][sp & 3], Dy);
                        }
                    }
#pragma unroll
                    for (int sp = 0; sp < KSA; ++sp)
                        twa[sp] = twn[sp];
                    // stage: left columns ascending, mirrored columns (W - w) ascending == w descending
                    const f32x4 yl = Ey - Dy;
                    const f32x4 yr = Ey + Dy;
                    *reinterpret_cast<f32x4*>(stL + r16 * STG_RS + 16 * t4 + 4 * kk) = yl;
                    *reinterpret_cast<f32x4*>(stR + r16 * STG_RS + 60 - 16 * t4 - 4 * kk) = f32x4{yr[3], yr[2], yr[1], yr[0]};
                }
            }
            // write the chunk out: pass q covers rows 4q..4q+3, each 16-lane group one 256-byte row segment
            const int c0 = STG_COLS * ch;                   // left chunk = columns c0 .. c0+63
            const int cr0 = W - c0 - (STG_COLS - 1);        // mirrored chunk = columns cr0 .. cr0+63  (= W - w)
            // fast path (wave-uniform): all 16 rows exist and both 64-column windows lie strictly inside their halves
            const bool rows_full = 16 * rt + 15 < H;
            const bool left_full = c0 + STG_COLS - 1 <= Wh;
            const bool right_full = cr0 > Wh && cr0 + STG_COLS - 1 < W;
            if (BF16 && !(UNO_ABLATE & 2) && rows_full && (left_full || right_full)) {
                // bf16 images: 8 columns per lane -> 16-byte stores, 8 rows per pass (half the store instructions of the
                // 4-column mapping below, which moved 8 bytes per lane)
                if constexpr (BF16) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int row = 8 * q + (lane >> 3), c8 = 8 * (lane & 7);
                        out_t* rowp = img + (size_t)(16 * rt + row) * W;
                        if (left_full) {
                            const f32x4 a = *reinterpret_cast<const f32x4*>(stL + row * STG_RS + c8);
                            const f32x4 b = *reinterpret_cast<const f32x4*>(stL + row * STG_RS + c8 + 4);
                            io_store8(rowp + c0 + c8, a, b);
                        }
                        if (right_full) {
                            const f32x4 a = *reinterpret_cast<const f32x4*>(stR + row * STG_RS + c8);
                            const f32x4 b = *reinterpret_cast<const f32x4*>(stR + row * STG_RS + c8 + 4);
                            io_store8(rowp + cr0 + c8, a, b);
                        }
                    }
                }
            } else if (!(UNO_ABLATE & 2) && rows_full && (left_full || right_full)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 4 * q + kk;
                    out_t* rowp = img + (size_t)(16 * rt + row) * W;
                    if (left_full) {
                        const f32x4 vl = *reinterpret_cast<const f32x4*>(stL + row * STG_RS + 4 * r16);
                        if (UNO_ABLATE & 4)      // timing probe: the row segment snapped to a 64-byte boundary (wrong place)
                            *reinterpret_cast<f32x4*>((reinterpret_cast<uintptr_t>(rowp + c0) & ~uintptr_t(63)) + 16 * r16) = vl;
                        else
                        io_store4(rowp + c0 + 4 * r16, vl[0], vl[1], vl[2], vl[3]);
                    }
                    if (right_full) {
                        const f32x4 vr = *reinterpret_cast<const f32x4*>(stR + row * STG_RS + 4 * r16);
                        if (UNO_ABLATE & 4)
                            *reinterpret_cast<f32x4*>((reinterpret_cast<uintptr_t>(rowp + cr0) & ~uintptr_t(63)) + 16 * r16) = vr;
                        else
                        io_store4(rowp + cr0 + 4 * r16, vr[0], vr[1], vr[2], vr[3]);
                    }
                }
            }
            const bool do_left = !(rows_full && left_full), do_right = !(rows_full && right_full);
            if (!(UNO_ABLATE & 2) && (do_left || do_right)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 4 * q + kk;
                    const int h = 16 * rt + row;
                    const f32x4 vl = *reinterpret_cast<const f32x4*>(stL + row * STG_RS + 4 * r16);
                    const f32x4 vr = *reinterpret_cast<const f32x4*>(stR + row * STG_RS + 4 * r16);
                    if (UNO_ABLATE & 32) {      // timing probe: LDS read-back kept, global stores dropped
                        asm volatile("" ::"v"(vl[0]), "v"(vl[3]), "v"(vr[0]), "v"(vr[3]));
                    } else if (h < H) {
                        out_t* rowp = img + (size_t)h * W;
                        const int cl = c0 + 4 * r16;        // first of this lane's four left columns
                        if (!do_left) {
                        } else if (cl + 3 <= Wh) {
                            io_store4(rowp + cl, vl[0], vl[1], vl[2], vl[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (cl + e <= Wh) io_store1(rowp + cl + e, vl[e]);
                        }
                        const int cr = cr0 + 4 * r16;       // mirrored columns must stay in (Wh, W-1]
                        if (!do_right) {
                        } else if (cr > Wh && cr + 3 < W) {
                            io_store4(rowp + cr, vr[0], vr[1], vr[2], vr[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (cr + e > Wh && cr + e < W) io_store1(rowp + cr + e, vr[e]);
                        }
                    }
                }
            }
        }
    }
}

template <int KS, int JT, bool BF16>
static int launch_inv_b(const Dft2dParams& p, hipStream_t s) {
    const int nrt = (p.H + 15) / 16;
    const int NW = (long long)p.H * p.W < 4096 ? 1 : pick_waves_per_image(nrt);     // small images (3-D planes): one wave each, more images in flight per CU
    const size_t lds = (size_t)(p.W + p.H) * sizeof(float2) + (size_t)NW * 2 * 16 * STG_RS * sizeof(float) + (size_t)KS * 64 * 4;
    if (lds > 160 * 1024) { set_error("dft2d_inv: grid %dx%d needs %zu B of LDS", p.H, p.W, lds); return -3; }
    auto k = dft2d_inv_kernel<KS, JT, BF16>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("dft2d_inv: cannot raise dynamic LDS to %zu", lds);
            return -4;
        }
    }
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_inv_kernel<%d, %d%s>", KS, JT, BF16 ? ", bf16" : "");
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * (BF16 ? 2.0 : 4.0) + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3(p.n_img), dim3(64 * NW), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_inv launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

template <int KS, int JT>
static int launch_inv_t(const Dft2dParams& p, hipStream_t s) {
    return p.bf16 ? launch_inv_b<KS, JT, true>(p, s) : launch_inv_b<KS, JT, false>(p, s);
}

#ifndef UNO_INV_KS_LO
#define UNO_INV_KS_LO 1
#define UNO_INV_KS_HI 12
#define UNO_INV_DISPATCH launch_dft2d_inv
#endif

// one translation unit instantiates a range of KS (the build compiles dft2d_inv.hip twice with different ranges)
int UNO_INV_DISPATCH(const Dft2dParams& p, hipStream_t s) {
    const int KS = (p.m2 + 3) / 4, JT = (2 * p.m1 + 15) / 16;
#define UNO_CASE(ks, jt) if (ks >= UNO_INV_KS_LO && ks <= UNO_INV_KS_HI && KS == ks && JT == jt) return launch_inv_t<ks, jt>(p, s);
#define UNO_ROW(ks) UNO_CASE(ks, 1) UNO_CASE(ks, 2) UNO_CASE(ks, 3) UNO_CASE(ks, 4) UNO_CASE(ks, 5)
    UNO_ROW(1) UNO_ROW(2) UNO_ROW(3) UNO_ROW(4) UNO_ROW(5) UNO_ROW(6)
    UNO_ROW(7) UNO_ROW(8) UNO_ROW(9) UNO_ROW(10) UNO_ROW(11) UNO_ROW(12)
#undef UNO_ROW
#undef UNO_CASE
    set_error("dft2d_inv: modes (%d, %d) exceed the compiled range (modes1 <= 40, modes2 <= 48)", p.m1, p.m2);
    return -2;
}

}  // namespace uno
