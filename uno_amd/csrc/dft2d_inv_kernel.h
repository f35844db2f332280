// K3 - pruned inverse 2-D DFT:  O (n_img, 2*m1, m2) c64  ->  y (n_img, H, W) f32
//
//   y[h][w] = Re sum_{j,l} scale * c_l * keep_j * O[j][l] * exp(+2 pi i (K_j h / H + l w / W))
//
// i.e. torch.fft.irfft2(out_ft, s=(H, W), norm="forward") of a spectrum that is zero outside the two
// low-frequency corners (reference integral_operators.py:190-206) - the zero-filled out_ft is never
// materialised.  With herm=0, mask=0 and scale=1/(H W) it is the gx stage of the backward pass.
//
// One workgroup = G images x NW waves per image (12 waves when the images are plentiful: the whole CU), one wave per
// 16-row tile of its image's output.
//   stage B' (columns): U^T[l][h] = sum_j O[l][j] exp(+i theta(j,h)):  M = modes, N = the tile's 16
//     rows, K = corner rows.  The O operand is loaded once per image into registers (A operand); the
//     mode <-> M-row assignment is permuted (row 4g+r computes mode 4r+g) so that ...
//   stage A' (rows): ... the stage-B' accumulators are directly the B operand of the row transform
//     (k-step s, lane group kk <-> mode 4s+kk), with the twiddles as A operand: D[w][h], i.e. every lane
//     ends up with FOUR CONSECUTIVE COLUMNS of one row.  Symmetric form: Ey = sum Ur cos, Dy = sum Ui sin
//     over w <= W/2, then y[w] = Ey - Dy and y[W-w] = Ey + Dy: half the flops, no padding waste in K.
//   twiddle operand of stage A' (TAB): cos / sin(2 pi l w / W) in MFMA operand layout [column tile][k-step][lane] is the
//     same for every row tile of every image, so the workgroup tabulates it once in LDS (35 KB at 421 columns x 20
//     modes) and the inner loop is ds_read_b64 at immediate offsets: no index walk (3 VALU per operand), no gather from
//     a W-entry table with its bank conflicts.  That table is what makes the workgroup span several images: it is
//     shared by all 12 waves.  Grids whose table does not fit (1024 columns x 32 modes) walk the W-entry table instead.
//   stores: a scattered 16-rows-per-instruction store pattern costs more than all the MFMAs (measured:
//     117 of 295 us), so each wave stages 64-column chunks of its 16 rows (left half and mirrored half)
//     in a private LDS buffer (ds_write_b128) and writes them out as 256-byte row segments with
//     16-byte-per-lane stores.
#pragma once
#include "uno_common.h"
#include <algorithm>
#include <cstdio>

namespace uno {

constexpr int STG_COLS = 64;            // columns per staged chunk (4 MFMA column tiles)
constexpr int STG_RS = STG_COLS + 4;    // LDS row stride in floats: 16-byte aligned, 4-bank skew per row
constexpr int INV_MAX_WAVES = 12;       // waves per workgroup (3 per SIMD: the register budget of the large instantiations)
constexpr size_t INV_LDS_BUDGET = 160 * 1024 - 2048;

// BF16: the images are written as bfloat16 (round to nearest even; config C5), everything before the store is f32.
// TAB: stage-A' twiddles come from the operand-layout table in LDS.
template <int KS, int JT, bool BF16, bool TAB>
__global__ __launch_bounds__(64 * INV_MAX_WAVES) void dft2d_inv_kernel(Dft2dParams p) {
    using out_t = typename IoElem<BF16>::type;
    constexpr int NT = (KS + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int NWT = nthreads >> 6;              // waves in the workgroup
    const int NW = p.nw;                        // waves per image
    const int Wh = W >> 1;                      // columns 0..Wh are computed, Wh+1..W-1 are their mirror images
    const int nwt = (Wh + 16) >> 4;             // 16-column tiles covering 0..Wh
    float* sStage = reinterpret_cast<float*>(smem);                       // [NWT][2][16][STG_RS]
    float2* sTwH = reinterpret_cast<float2*>(sStage + NWT * 2 * 16 * STG_RS);
    float2* sTabA = sTwH + H;                                             // TAB: [nwt][KS][64] | else: sTwW [W] + sIdxA0 [KS][64]
    float2* sTwW = sTabA;
    unsigned* sIdxA0 = reinterpret_cast<unsigned*>(sTwW + W);

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const unsigned W8 = 8u * W, H8 = 8u * H;

    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];
    if constexpr (TAB) {
        // entry (wt, sp, lane = (i, kq)): column w = 16 wt + i, mode l = 4 sp + kq  (modes >= m2 face zero operands)
        for (int e = tid; e < nwt * KS * 64; e += nthreads) {
            const int ln = e & 63, q = e >> 6;
            const int sp = q % KS, wt = q / KS;
            const unsigned l = (unsigned)min(4 * sp + (ln >> 4), m2 - 1);
            const unsigned w = (unsigned)(16 * wt + (ln & 15));
            sTabA[e] = p.twW[(l * w) % (unsigned)W];
        }
    } else {
        for (int n = tid; n < W; n += nthreads) sTwW[n] = p.twW[n];
        for (int e = tid; e < KS * 64; e += nthreads) {
            const unsigned l = (unsigned)min(4 * (e >> 6) + ((e & 63) >> 4), m2 - 1);
            sIdxA0[e] = 8u * ((l * (unsigned)(e & 15)) % (unsigned)W);
        }
    }
    __syncthreads();

    const int slot = wave / NW, wsub = wave - slot * NW;
    const int image = sweep_x(p.rev) * (NWT / NW) + slot;
    if (image >= p.n_img) return;               // no barrier below

    // stage-B' A operand.  The corner rows come in +-k pairs (lo corner row k <-> frequency +k, hi corner row
    // 2 m1 - k <-> frequency -k), so with P_k = O[+k] + O[-k], M_k = O[+k] - O[-k]:
    //     U[h] = sum_{k=0}^{m1} cos(theta_k h) P_k + i sin(theta_k h) M_k          (theta_k = 2 pi k / H)
    // i.e. m1 + 1 real twiddle pairs instead of 2 m1 complex ones - 40 % fewer MFMAs in this stage.
    // Operand lane (rho = r16 -> mode 16 t + 4 (rho & 3) + (rho >> 2), k-slot kk -> k = 4 ks + kk).
    constexpr int KSK = 2 * JT + 1;                       // >= ceil((m1 + 1) / 4)
    const int ksk = (m1 + 4) >> 2;                        // k-steps actually needed
    const float2* O = reinterpret_cast<const float2*>(p.in) + spectrum_index(p, image) * 2 * m1 * m2;
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
    // stage-A' twiddle walk (A operand) of the table-less form: lane (i = r16 -> column w = 16 wt + r16, k-slot kk -> mode 4 sp + kk)
    unsigned stepA[TAB ? 1 : KS];
    if constexpr (!TAB) {
#pragma unroll
        for (int sp = 0; sp < KS; ++sp) {
            const unsigned l = (unsigned)min(4 * sp + kk, m2 - 1);
            stepA[sp] = 8u * ((16u * l) % (unsigned)W);
        }
    }

    out_t* img = reinterpret_cast<out_t*>(p.out) + (size_t)image * H * W;
    const int nrt = (H + 15) >> 4;
    const int nchunk = (nwt + 3) >> 2;
    float* stL = sStage + (size_t)wave * 2 * 16 * STG_RS;     // this wave's left-half chunk  [16][STG_RS]
    float* stR = stL + 16 * STG_RS;                            // mirrored-half chunk
    const float2* tabLane = sTabA + lane;

    for (int rt = wsub; rt < nrt; rt += NW) {
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
        unsigned idxA[TAB ? 1 : KS];
        float2 twa[KS];
        asm volatile("" ::: "memory");          // keep the loop-invariant table reads below inside the loop (register pressure)
        if constexpr (TAB) {
#pragma unroll
            for (int sp = 0; sp < KS; ++sp) twa[sp] = tabLane[sp * 64];
        } else {
#pragma unroll
            for (int sp = 0; sp < KS; ++sp) {
                const unsigned i0 = sIdxA0[sp * 64 + lane];
                twa[sp] = lds_tw(sTwW, i0);
                idxA[sp] = wrap_add(i0, stepA[sp], W8);
            }
        }
        for (int ch = 0; ch < nchunk; ++ch) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int wt = 4 * ch + t4;
                if (wt < nwt) {
                    float2 twn[KS];
                    if constexpr (TAB) {
                        const float2* nxt = tabLane + (size_t)min(wt + 1, nwt - 1) * (KS * 64);
#pragma unroll
                        for (int sp = 0; sp < KS; ++sp) twn[sp] = nxt[sp * 64];
                    } else {
#pragma unroll
                        for (int sp = 0; sp < KS; ++sp) {
                            twn[sp] = lds_tw(sTwW, idxA[sp]);
                            idxA[sp] = wrap_add(idxA[sp], stepA[sp], W8);
                        }
                    }
                    f32x4 Ey = f32x4{0, 0, 0, 0}, Dy = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int sp = 0; sp < KS; ++sp) {
                        Ey = mfma16(twa[sp].x, Ur[sp >> 2][sp & 3], Ey);
                        Dy = mfma16(twa[sp].y, Ui[sp >> 2][sp & 3], Dy);
                    }
#pragma unroll
                    for (int sp = 0; sp < KS; ++sp) twa[sp] = twn[sp];
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
            if (BF16 && rows_full && (left_full || right_full)) {
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
            } else if (rows_full && (left_full || right_full)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 4 * q + kk;
                    out_t* rowp = img + (size_t)(16 * rt + row) * W;
                    if (left_full) {
                        const f32x4 vl = *reinterpret_cast<const f32x4*>(stL + row * STG_RS + 4 * r16);
                        io_store4(rowp + c0 + 4 * r16, vl[0], vl[1], vl[2], vl[3]);
                    }
                    if (right_full) {
                        const f32x4 vr = *reinterpret_cast<const f32x4*>(stR + row * STG_RS + 4 * r16);
                        io_store4(rowp + cr0 + 4 * r16, vr[0], vr[1], vr[2], vr[3]);
                    }
                }
            }
            const bool do_left = !(rows_full && left_full), do_right = !(rows_full && right_full);
            if (do_left || do_right) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 4 * q + kk;
                    const int h = 16 * rt + row;
                    const f32x4 vl = *reinterpret_cast<const f32x4*>(stL + row * STG_RS + 4 * r16);
                    const f32x4 vr = *reinterpret_cast<const f32x4*>(stR + row * STG_RS + 4 * r16);
                    if (h < H) {
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

// ---------------------------------------------------------------------------------------------------------------- K3, full-tile form
// What bounds the chunked kernel above is not arithmetic but the SHAPE of its stores (tools/probes/store_probe.hip, 1024
// images of 416 x 421 floats, no arithmetic at all): 256-byte row segments at 4-byte alignment stream at 3.1 TB/s, the same
// segments snapped to 64-byte boundaries at 4.0, and a wave that writes its 16-row tile as ONE contiguous run of whole,
// 128-byte-aligned cache lines at 5.6 TB/s.  Every partially written line costs about as much as a whole one.  Rows of an odd
// length start at 4-byte alignment, so whole lines straddle rows: this form builds the complete 16 x W tile in LDS (the tile
// is one contiguous 16 W float run of the image) at the same offset modulo 128 bytes as in memory and writes it out as 1 KB
// per instruction, whole lines only.  With one wave per image the partial line at the end of a tile is carried into the next
// tile's buffer, so an image is written with two partial lines in total.  27 KB of LDS per wave (421 columns) allow four waves
// per CU - one per SIMD, which is what the MFMA chains need when the stores are asynchronous; the kernel is then bound by HBM
// writes at the contiguous-stream rate.
template <int KS, int JT>
__global__ __launch_bounds__(256) void dft2d_inv_ft_kernel(Dft2dParams p) {
    constexpr int NT = (KS + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int NWT = nthreads >> 6;
    const int NW = p.nw;
    const int Wh = W >> 1;
    const int nwt = (Wh + 16) >> 4;
    float2* sTabA = reinterpret_cast<float2*>(smem);                      // [nwt][KS][64]
    float2* sTwH = sTabA + nwt * KS * 64;
    float* sTile = reinterpret_cast<float*>(sTwH + ((H + 1) & ~1));        // [NWT][tile_stride], 16-byte aligned
    const int tile_stride = (16 * W + 32 + 64 + 3) & ~3;                   // tile + alignment phase + one dump slot per lane

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const unsigned H8 = 8u * H;

    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];
    for (int e = tid; e < nwt * KS * 64; e += nthreads) {
        const int ln = e & 63, q = e >> 6;
        const int sp = q % KS, wt = q / KS;
        const unsigned l = (unsigned)min(4 * sp + (ln >> 4), m2 - 1);
        const unsigned w = (unsigned)(16 * wt + (ln & 15));
        sTabA[e] = p.twW[(l * w) % (unsigned)W];
    }
    __syncthreads();

    const int slot = wave / NW, wsub = wave - slot * NW;
    const int image = sweep_x(p.rev) * (NWT / NW) + slot;
    if (image >= p.n_img) return;               // no barrier below

    constexpr int KSK = 2 * JT + 1;
    const int ksk = (m1 + 4) >> 2;
    const float2* O = reinterpret_cast<const float2*>(p.in) + spectrum_index(p, image) * 2 * m1 * m2;
    float Pr[NT][KSK], Pi[NT][KSK], Mr[NT][KSK], Mi[NT][KSK];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int l = 16 * t + 4 * (r16 & 3) + (r16 >> 2);
        const float cs = p.scale * ((p.herm && l < m2) ? herm_weight(l, W) : 1.0f);
#pragma unroll
        for (int ks = 0; ks < KSK; ++ks) {
            const int k = 4 * ks + kk;
            float2 vp = make_float2(0.f, 0.f), vm = make_float2(0.f, 0.f);
            if (l < m2 && k < m1 && !(p.mask && !row_survives(k, m1, H))) vp = O[(size_t)k * m2 + l];
            if (l < m2 && k >= 1 && k <= m1) vm = O[(size_t)(2 * m1 - k) * m2 + l];
            Pr[t][ks] = (vp.x + vm.x) * cs; Pi[t][ks] = (vp.y + vm.y) * cs;
            Mr[t][ks] = (vp.x - vm.x) * cs; Mi[t][ks] = (vp.y - vm.y) * cs;
        }
    }

    float* img = p.out + (size_t)image * H * W;
    const int nrt = (H + 15) >> 4;
    float* buf = sTile + (size_t)wave * tile_stride;
    const int dump = 16 * W + 32 + lane;                    // where guarded-out elements go
    const float2* tabLane = sTabA + lane;
    const int wfast_hi = W - Wh - 1;                        // largest w whose mirror column W - w lies in the right half
    const bool chain = NW == 1;                             // consecutive tiles by one wave: carry the partial line

    // stage B' of row tile rt -> (Ur, Ui); `between(ks)` runs after the MFMAs of k-step ks (the store phase of the PREVIOUS tile is
    // threaded through here: its LDS reads and global stores issue in the shadow of the column stage's MFMAs)
    f32x4 Ur[NT], Ui[NT];
    auto stage_b = [&](int rt, auto&& between) {
        const unsigned hB = (unsigned)min(16 * rt + r16, H - 1);
        const unsigned a4 = 8u * ((4u * hB) % (unsigned)H);
        unsigned aj = 8u * (((unsigned)kk * hB) % (unsigned)H);
#pragma unroll
        for (int t = 0; t < NT; ++t) { Ur[t] = f32x4{0, 0, 0, 0}; Ui[t] = f32x4{0, 0, 0, 0}; }
        float2 twb = lds_tw(sTwH, aj);
#pragma unroll
        for (int ks = 0; ks < KSK; ++ks) {
            aj = wrap_add(aj, a4, H8);
            const float2 twn = lds_tw(sTwH, aj);
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
            between(ks);
        }
    };
    if (wsub < nrt) stage_b(wsub, [](int) {});

    for (int rt = wsub; rt < nrt; rt += NW) {
        float* tile = img + (size_t)rt * 16 * W;
        const int rows = min(16, H - 16 * rt);
        const int phase = (int)((reinterpret_cast<uintptr_t>(tile) >> 2) & 31);       // LDS index == memory offset (mod 128 B)

        // ---- stage A' into the LDS tile: lane (h = r16, g = kk) owns columns 16 wt + 4 g + e and their mirrors W - (...).
        // One wave per SIMD: the MFMA pipe only stays busy if nothing in the instruction stream waits for the MFMAs just
        // issued, so the results of a column tile are staged while the NEXT tile's chain runs (two accumulator sets,
        // ping-pong, no copies), and interior tiles take a branch-free path.
        const int rowbase = phase + r16 * W;
        const bool row_ok = r16 < rows;
        // twiddles of the column tiles in two sets, ping-pong: even tiles multiply out of twA while twB is filled, odd tiles the other
        // way round (round 5: the single set was refilled by 2 KS copies after every chain, which no MFMA covered)
        float2 twA[KS], twB[KS];
#pragma unroll
        for (int sp = 0; sp < KS; ++sp) twA[sp] = tabLane[sp * 64];
        auto chain_sets = [&](int wt, f32x4& Ey, f32x4& Dy, float2 (&cur)[KS], float2 (&nxtset)[KS]) {
            // request the NEXT tile's twiddles first (sched_barrier keeps the reads in front of the MFMAs), then run this chain
            const float2* nxt = tabLane + (size_t)min(wt + 1, nwt - 1) * (KS * 64);
#pragma unroll
            for (int sp = 0; sp < KS; ++sp) nxtset[sp] = nxt[sp * 64];
            __builtin_amdgcn_sched_barrier(0);
            Ey = f32x4{0, 0, 0, 0}; Dy = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int sp = 0; sp < KS; ++sp) {
                Ey = mfma16(cur[sp].x, Ur[sp >> 2][sp & 3], Ey);
                Dy = mfma16(cur[sp].y, Ui[sp >> 2][sp & 3], Dy);
            }
        };
        auto chain_mfma = [&](int wt, f32x4& Ey, f32x4& Dy) {
            if (wt & 1) chain_sets(wt, Ey, Dy, twB, twA); else chain_sets(wt, Ey, Dy, twA, twB);      // (uniform)
        };
        auto stage_fast = [&](int wt, const f32x4& Ey, const f32x4& Dy) {
            const int w0 = 16 * wt + 4 * kk;
            const f32x4 yl = Ey - Dy;
            const f32x4 yr = Ey + Dy;
            float* pl = buf + rowbase + w0;
            float* pr = buf + rowbase + W - w0 - 3;
            pl[0] = yl[0]; pl[1] = yl[1]; pl[2] = yl[2]; pl[3] = yl[3];
            pr[3] = yr[0]; pr[2] = yr[1]; pr[1] = yr[2]; pr[0] = yr[3];
        };
        auto stage_guarded = [&](int wt, const f32x4& Ey, const f32x4& Dy) {
            const int w0 = 16 * wt + 4 * kk;
            const f32x4 yl = Ey - Dy;
            const f32x4 yr = Ey + Dy;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int w = w0 + e;
                buf[(row_ok && w <= Wh) ? rowbase + w : dump] = yl[e];
                buf[(row_ok && w >= 1 && w <= wfast_hi) ? rowbase + W - w : dump] = yr[e];
            }
        };
        // column tiles [1, nfast) need no guards when all 16 rows exist
        const int nfast = rows == 16 ? max(1, min(nwt, (wfast_hi + 1) >> 4)) : 1;
        f32x4 E0, D0, E1, D1;
        chain_mfma(0, E0, D0);
        int wt = 1;
        for (; wt + 1 < nfast; wt += 2) {
            chain_mfma(wt, E1, D1);
            if (wt == 1) stage_guarded(0, E0, D0); else stage_fast(wt - 1, E0, D0);
            chain_mfma(wt + 1, E0, D0);
            stage_fast(wt, E1, D1);
        }
        // tail: the tile held in (E0, D0) is wt - 1; the remaining tiles take the guarded path
        for (; wt < nwt; ++wt) {
            chain_mfma(wt, E1, D1);
            if (wt - 1 >= 1 && wt - 1 < nfast) stage_fast(wt - 1, E0, D0); else stage_guarded(wt - 1, E0, D0);
            E0 = E1; D0 = D1;
        }
        if (nwt - 1 >= 1 && nwt - 1 < nfast) stage_fast(nwt - 1, E0, D0); else stage_guarded(nwt - 1, E0, D0);

        // ---- the tile goes out as whole 128-byte lines: LDS index i <-> memory gbase[i]
        float* gbase = tile - phase;
        const int total = phase + rows * W;
        const bool first = !chain || rt == 0, last = !chain || rt + NW >= nrt;
        const int lo = first ? phase : 0;                   // chained tiles start with the carried head of the line
        const int hi = last ? total : (total & ~31);
        auto store_guarded = [&](int i) {
            if (i >= hi) return;
            if (i >= lo && i + 3 < hi) {
                *reinterpret_cast<f32x4*>(gbase + i) = *reinterpret_cast<const f32x4*>(buf + i);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i + e >= lo && i + e < hi) gbase[i + e] = buf[i + e];
            }
        };
        const int nfull = hi >> 8;                          // instructions [1, nfull) cover whole 1 KB runs inside [lo, hi)
        store_guarded(4 * lane);
        int it = 1;
        auto store_batch = [&]() {                          // four whole 1 KB runs: reads first, then the stores
            if (it + 4 <= nfull) {
                const float* src = buf + 256 * it + 4 * lane;
                float* dst = gbase + 256 * it + 4 * lane;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 256);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + 512);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + 768);
                // plain stores: the non-temporal form measured 161 -> 157.5 us (noise) and would keep the block's output out of
                // the Infinity Cache for its consumer
                *reinterpret_cast<f32x4*>(dst) = v0;
                *reinterpret_cast<f32x4*>(dst + 256) = v1;
                *reinterpret_cast<f32x4*>(dst + 512) = v2;
                *reinterpret_cast<f32x4*>(dst + 768) = v3;
                it += 4;
            }
        };
        // the next tile's column stage does not touch the LDS tile: its MFMAs cover this tile's store phase
        if (rt + NW < nrt) stage_b(rt + NW, [&](int) { store_batch(); });
        while (it + 4 <= nfull) store_batch();
        for (; it < nfull; ++it)
            *reinterpret_cast<f32x4*>(gbase + 256 * it + 4 * lane) = *reinterpret_cast<const f32x4*>(buf + 256 * it + 4 * lane);
        if (nfull >= 1) store_guarded(256 * nfull + 4 * lane);
        if (!last) {
            // carry the partial last line to the head of the next tile's image (same buffer, next phase = total & 31)
            const int rem = total & 31;
            float v = 0.f;
            if (lane < rem) v = buf[(total & ~31) + lane];
            if (lane < rem) buf[lane] = v;
        }
    }
}

// Workgroup geometry: NW waves per image, G images per workgroup (NW * G <= 12 waves).  All workgroups take the same time, so
// the launch runs in ceil(groups / (CUs * workgroups per CU)) rounds of ceil(row tiles / NW) tile times each; choose the
// (NW, G) with the fewest tile times, more waves per image on a tie.
struct InvGeometry { int nw, g; size_t lds; bool tab; };

static int device_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return usable_cus(cus);
}

static size_t inv_lds_bytes(const Dft2dParams& p, int KS, int waves, bool tab) {
    const int nwt = ((p.W >> 1) + 16) >> 4;
    const size_t common = (size_t)waves * 2 * 16 * STG_RS * sizeof(float) + (size_t)p.H * sizeof(float2);
    return common + (tab ? (size_t)nwt * KS * 64 * sizeof(float2) : (size_t)p.W * sizeof(float2) + (size_t)KS * 64 * 4);
}

static InvGeometry inv_geometry(const Dft2dParams& p, int KS, bool allow_tab) {
    const int nrt = (p.H + 15) / 16, cus = device_cu_count();
    InvGeometry best{1, 1, 0, false};
    long long best_cost = -1;
    for (int nw = 1; nw <= 4 && nw <= nrt; ++nw) {
        int g = INV_MAX_WAVES / nw;
        // few images: keep at least one workgroup per CU before stacking images into a workgroup
        while (g > 1 && (long long)(p.n_img + g - 1) / g < cus) --g;
        const bool tab = allow_tab && inv_lds_bytes(p, KS, nw * g, true) <= INV_LDS_BUDGET;
        const size_t lds = inv_lds_bytes(p, KS, nw * g, tab);
        if (lds > INV_LDS_BUDGET) continue;
        const long long groups = (p.n_img + g - 1) / g;
        const long long per_cu = std::max<long long>(1, std::min<long long>((long long)(INV_LDS_BUDGET / lds), INV_MAX_WAVES / (nw * g)));
        const long long rounds = (groups + cus * per_cu - 1) / (cus * per_cu);
        const long long cost = rounds * ((nrt + nw - 1) / nw) * 64 + (tab ? 0 : 8);       // table-less form is slower per tile
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = InvGeometry{nw, g, lds, tab}; }
    }
    if (best_cost < 0) best.lds = inv_lds_bytes(p, KS, 1, false);
    return best;
}

template <int KS, int JT, bool BF16, bool TAB>
static int launch_inv_k(Dft2dParams p, const InvGeometry& g, hipStream_t s) {
    auto k = dft2d_inv_kernel<KS, JT, BF16, TAB>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), g.lds, lds_slot)) { set_error("dft2d_inv: cannot raise dynamic LDS to %zu", g.lds); return -4; }
    p.nw = g.nw;
    p.rev = next_sweep_reversed(SWEEP_K3);
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_inv_kernel<%d, %d, %s, %s>", KS, JT, BF16 ? "true" : "false", TAB ? "true" : "false");
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * (BF16 ? 2.0 : 4.0) + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3((p.n_img + g.g - 1) / g.g), dim3(64 * g.nw * g.g), g.lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_inv launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

// full-tile form: up to 4 waves per workgroup (NW per image x G images), whole tiles in LDS
static size_t inv_ft_lds_bytes(const Dft2dParams& p, int KS, int waves) {
    const int nwt = ((p.W >> 1) + 16) >> 4;
    const size_t tile_stride = (size_t)((16 * p.W + 32 + 64 + 3) & ~3);
    return (size_t)nwt * KS * 64 * sizeof(float2) + (size_t)((p.H + 1) & ~1) * sizeof(float2) + (size_t)waves * tile_stride * sizeof(float);
}

static bool inv_ft_geometry(const Dft2dParams& p, int KS, InvGeometry* out) {
    // the staging writes walk 16 rows at a stride of W floats: W % 8 == 0 puts them on 4 or fewer LDS banks
    if (p.bf16 || p.W % 8 == 0 || p.W < 16) return false;
    const int nrt = (p.H + 15) / 16, cus = device_cu_count();
    long long best_cost = -1;
    for (int nw = 1; nw <= 4 && nw <= nrt; nw *= 2) {
        int g = 4 / nw;
        while (g > 1 && (long long)(p.n_img + g - 1) / g < cus) --g;
        const size_t lds = inv_ft_lds_bytes(p, KS, nw * g);
        if (lds > INV_LDS_BUDGET) continue;
        // a CU should hold at least four waves (one per SIMD)
        const long long per_cu = std::max<long long>(1, std::min<long long>((long long)(INV_LDS_BUDGET / lds), 16 / (nw * g)));
        if (per_cu * nw * g < 4 && (long long)p.n_img * nw >= 4LL * cus) continue;
        const long long groups = (p.n_img + g - 1) / g;
        const long long rounds = (groups + cus * per_cu - 1) / (cus * per_cu);
        const long long cost = rounds * ((nrt + nw - 1) / nw);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; *out = InvGeometry{nw, g, lds, true}; }
    }
    return best_cost >= 0;
}

template <int KS, int JT>
static int launch_inv_ft(Dft2dParams p, const InvGeometry& g, hipStream_t s) {
    auto k = dft2d_inv_ft_kernel<KS, JT>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), g.lds, lds_slot)) { set_error("dft2d_inv: cannot raise dynamic LDS to %zu", g.lds); return -4; }
    p.nw = g.nw;
    p.rev = next_sweep_reversed(SWEEP_K3);
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_inv_ft_kernel<%d, %d>", KS, JT);
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * 4.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3((p.n_img + g.g - 1) / g.g), dim3(64 * g.nw * g.g), g.lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_inv launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

template <int KS, int JT>
static int launch_inv_t(const Dft2dParams& p, hipStream_t s) {
    InvGeometry ft;
    if (inv_ft_geometry(p, KS, &ft)) return launch_inv_ft<KS, JT>(p, ft, s);
    const InvGeometry g = inv_geometry(p, KS, !p.bf16);
    if (g.lds == 0 || g.lds > INV_LDS_BUDGET) { set_error("dft2d_inv: grid %dx%d needs %zu B of LDS", p.H, p.W, g.lds); return -3; }
    if (p.bf16) return launch_inv_k<KS, JT, true, false>(p, g, s);
    return g.tab ? launch_inv_k<KS, JT, false, true>(p, g, s) : launch_inv_k<KS, JT, false, false>(p, g, s);
}

// A translation unit instantiates the k-step counts [LO, HI] (dft2d_inv.hip / _b / _c split the range: 180 kernels otherwise
// compile for minutes in one unit).
template <int LO, int HI>
static int dispatch_inv_range(const Dft2dParams& p, hipStream_t s) {
    const int KS = (p.m2 + 3) / 4, JT = (2 * p.m1 + 15) / 16;
#define UNO_CASE(ks, jt) if constexpr (ks >= LO && ks <= HI) { if (KS == ks && JT == jt) return launch_inv_t<ks, jt>(p, s); }
#define UNO_ROW(ks) UNO_CASE(ks, 1) UNO_CASE(ks, 2) UNO_CASE(ks, 3) UNO_CASE(ks, 4) UNO_CASE(ks, 5)
    UNO_ROW(1) UNO_ROW(2) UNO_ROW(3) UNO_ROW(4) UNO_ROW(5) UNO_ROW(6)
    UNO_ROW(7) UNO_ROW(8) UNO_ROW(9) UNO_ROW(10) UNO_ROW(11) UNO_ROW(12)
#undef UNO_ROW
#undef UNO_CASE
    set_error("dft2d_inv: modes (%d, %d) outside the compiled range of this unit", p.m1, p.m2);
    return -2;
}

}  // namespace uno
