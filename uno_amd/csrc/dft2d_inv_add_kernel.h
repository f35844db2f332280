// K3-A - K3 (full-tile form) with the UP-SAMPLED point-wise branch added in registers before the tile is staged:
//
//   y[h][w] = Re iDFT_trunc(O)[h][w]  +  sum_{u,v} Rh[h][u] Rw[w][v] t[u][v]
//
// i.e. `x1_out + x2_out` of an up-sampling operator block (reference integral_operators.py:272-273, with x2_out =
// F.interpolate(conv(x), bicubic, align_corners, antialias), :240-242, evaluated from the LOW-resolution 1x1-convolution result t)
// and, transposed, the input gradient of a down-sampling block.  Rounds 1-5 ran this as K3 (writes y) followed by the accumulating
// form of K7 (reads t, reads y, writes y): 1.83 GB for an ideal 1.02 GB at the 223^2 -> 446^2 level of the Darcy model, the largest
// family of the step.  Here y is written once and t is read (twice, from L2) by the kernel that produces y.
//
// Why this form and not the one costed in round 2 (~1 700 VALU / LDS instructions per tile, LDS full at 446 columns): the kernel is
// bound by its stores with the MFMA pipe ~40 % busy, LDS is full (4 x 28.9 KB of tiles + 35.8 KB twiddle operands of 160 KB), one wave
// per SIMD leaves nothing to hide VALU work behind - but MFMA slots are free.  Both 1-D operators are banded: a 16-row output tile
// reads <= 12 rows of t, 16 output columns read <= 12 columns (up-sampling by two or more).  So per (column tile, side):
//   stage 1'  C^T[u][w] = sum_v t[u][v] Rw[w][v]      A operand = t straight from global: lane (row u, k-slot kk) loads the 12-byte piece
//                                                     t[u][v0 + 3 kk .. + 2] - ONE load per lane, k-step ks <-> column v0 + 3 kk + ks;
//                                                     B operand = the column operator in operand layout, loop-invariant: registers
//   stage 2'  D[w][h]  += sum_u C^T[u][w] Rh[h][u]    the stage-1' accumulators ARE the A operand (register e of lane (kk, n) = row
//                                                     u = 3 kk + e of C^T, column n: the trick K3's own two stages use);
//                                                     B operand = the row operator of the tile: three registers per tile
// D[w][h] comes out in the layout of the transform's own result (lane (h, g): columns 4 g .. 4 g + 3 of the column tile and, for the
// mirrored side, W - (4 g ..)), so the addend meets the transform in registers: 6 + 6 MFMAs per column tile next to the transform's
// 2 KS, no LDS, ~4 VALU.
// Memory ordering: loads and stores share the vmcnt counter, and this kernel lives on stores that drain asynchronously over a whole
// tile time - waiting for a load means waiting for every store issued before it (knock-outs, tools/dev/k3a_time.py: with the loads
// requested AFTER the column loop the kernel took 397 us, 267 without the loads, 187 without loads and addend MFMAs = K3 alone).  So the
// loads of tile rt + 1 (28 pieces of t + 3 operator registers) are issued at the TOP of tile rt, straight behind the stores of tile
// rt - 1, and consumed after tile rt's column loop (stage 1' of every column tile, results parked in registers) but before tile rt's
// stores are issued: what they wait for has had a whole column loop to drain.  The column loop is fully unrolled (NWTM column tiles) so
// that the parked operands are registers.
#pragma once
#include "dft2d_inv_kernel.h"
#include <type_traits>

namespace uno {

constexpr int ADD_KE = 3;           // k-steps of both banded operators: bands of <= 12 source rows / columns per 16 outputs

__device__ __forceinline__ f32x4 f4(float a, float b, float c, float d) { return f32x4{a, b, c, d}; }
typedef float f32x3 __attribute__((ext_vector_type(3)));

template <int KS, int KSK, int NWTM>
__global__ __launch_bounds__(256, ((NWTM <= 7 && KS <= 4) ? 2 : 1)) void dft2d_inv_ft_add_kernel(Dft2dParams p) {
    constexpr int NT = (KS + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int NWT = nthreads >> 6;
    const int NW = p.nw;
    const int Wh = W >> 1;
    constexpr int nwt = NWTM;                                              // == (Wh + 16) >> 4 (launcher): the column loop is straight-line code
    float2* sTabA = reinterpret_cast<float2*>(smem);                      // [nwt][KS][64]
    float2* sTwH = sTabA + nwt * KS * 64;
    float* sTile = reinterpret_cast<float*>(sTwH + ((H + 1) & ~1));
    const int tile_stride = (16 * W + 32 + 64 + 3) & ~3;

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

    const int ko = p.exp & 0xff;                 // development knock-outs: 1 no addend loads, 2 no spectrum reload, 4 no stores, 8 no stage 1', 16 no stage 2'
    // KSK == (m1 + 4) >> 2 k-steps of the column stage exactly (K3 compiles 2 JT + 1 >= that and skips the rest at run time: here every
    // register counts)
    constexpr int ksk = KSK;
    constexpr bool EARLY = !(NWTM <= 7 && KS <= 4);      // one wave per SIMD: operands requested DURING the tile's column loop; two waves per SIMD: after it
    const float2* O = reinterpret_cast<const float2*>(p.in) + spectrum_index(p, image) * 2 * m1 * m2;

    // ---- the addend's loop-invariant operands: column operator, B operand of stage 1' (lane (k-slot kk, column n = r16))
    const int Hs = p.add_Hs, Ws = p.add_Ws;
    float colB[NWTM][2][ADD_KE];
#pragma unroll
    for (int wt = 0; wt < NWTM; ++wt)
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int ks = 0; ks < ADD_KE; ++ks)
                colB[wt][sd][ks] = p.add_colop[((size_t)(wt * 2 + sd) * ADD_KE + ks) * 64 + lane];
    // t of this image through a buffer resource: reads past the image return zero (they only ever meet zero weights)
    const float* timg = p.add_src + (size_t)image * Hs * Ws;
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(timg), 0, Hs * Ws * 4, 0x00020000);
    // A-operand row of this lane: MFMA row m = r16 = 4 a + e <-> source row p0 + 3 a + e (e < 3; e = 3: unused, any valid row)
    const int urel = ADD_KE * (r16 >> 2) + min(r16 & 3, ADD_KE - 1);

    float* img = p.out + (size_t)image * H * W;
    const int nrt = (H + 15) >> 4;
    float* buf = sTile + (size_t)wave * tile_stride;
    const int dump = 16 * W + 32 + lane;
    const float2* tabLane = sTabA + lane;
    const int wfast_hi = W - Wh - 1;
    const bool chain = NW == 1;

    // column stage's A operand: P_k = O[+k] + O[-k], M_k = O[+k] - O[-k] (dft2d_inv_kernel.h), once per image
    float Pr[NT][KSK], Pi[NT][KSK], Mr[NT][KSK], Mi[NT][KSK];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int l = 16 * t + 4 * (r16 & 3) + (r16 >> 2);
        const float cs = p.scale * ((p.herm && l < m2) ? herm_weight(l, W) : 1.0f);
#pragma unroll
        for (int ks = 0; ks < KSK; ++ks) {
            const int k = 4 * ks + kk;
            float2 vp = make_float2(0.f, 0.f), vm = make_float2(0.f, 0.f);
            if (!(ko & 2) && l < m2 && k < m1 && !(p.mask && !row_survives(k, m1, H))) vp = O[(size_t)k * m2 + l];
            if (!(ko & 2) && l < m2 && k >= 1 && k <= m1) vm = O[(size_t)(2 * m1 - k) * m2 + l];
            Pr[t][ks] = (vp.x + vm.x) * cs; Pi[t][ks] = (vp.y + vm.y) * cs;
            Mr[t][ks] = (vp.x - vm.x) * cs; Mi[t][ks] = (vp.y - vm.y) * cs;
        }
    }
    f32x4 Ur[NT], Ui[NT];
    auto stage_b = [&](int rt) {
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
        }
    };
    // the addend of row tile rt, first half: operands requested (request), stage 1' of every column tile (reduce) -> Ct
    float rowB[ADD_KE], rowN[ADD_KE];           // row operator of the tile (B operand of stage 2') and of the next one
    f32x3 piece[NWTM][2];
    f32x4 Ct[NWTM][2];                          // C^T of (column tile, side): A operand of stage 2'
    // first source column of every (column tile, side), byte offset, once (scalar registers): a load of the table inside the tile loop
    // is no longer provably unclobbered and becomes a vector load + readfirstlane loop in the middle of the MFMA stream
    int v0s[NWTM][2];
#pragma unroll
    for (int wt = 0; wt < NWTM; ++wt)
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) v0s[wt][sd] = __builtin_amdgcn_readfirstlane(p.add_v0[wt * 2 + sd] * 4);
    int voff_next = 0;
    auto request_rows = [&](int rt) {            // row operator of tile rt + this lane's byte offset into t for its pieces
        if (ko & 1) {
#pragma unroll
            for (int e = 0; e < ADD_KE; ++e) rowN[e] = 0.f;
            return;
        }
#pragma unroll
        for (int e = 0; e < ADD_KE; ++e) rowN[e] = p.add_rowop[((size_t)rt * ADD_KE + e) * 64 + lane];
        const int p0 = __builtin_amdgcn_readfirstlane(p.add_p0[rt]);
        voff_next = (min(p0 + urel, Hs - 1) * Ws + ADD_KE * kk) * 4;
    };
    auto request_piece = [&](int wt) {           // the two 12-byte pieces of column tile wt (compile-time wt)
        if (ko & 1) { piece[wt][0] = f32x3{0, 0, 0}; piece[wt][1] = f32x3{0, 0, 0}; return; }
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            piece[wt][sd] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(trsrc, voff_next, v0s[wt][sd], 0));
        }
    };
    auto request = [&](int rt) {
        request_rows(rt);
#pragma unroll
        for (int wt = 0; wt < NWTM; ++wt) request_piece(wt);
    };
    auto reduce = [&]() {
#pragma unroll
        for (int e = 0; e < ADD_KE; ++e) rowB[e] = rowN[e];
        if (ko & 8) {
#pragma unroll
            for (int wt = 0; wt < NWTM; ++wt) { Ct[wt][0] = f32x4{0, 0, 0, 0}; Ct[wt][1] = f32x4{0, 0, 0, 0}; }
            return;
        }
#pragma unroll
        for (int wt = 0; wt < NWTM; ++wt) {
            f32x4 c0 = f32x4{0, 0, 0, 0}, c1 = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < ADD_KE; ++ks) {
                c0 = mfma16(piece[wt][0][ks], colB[wt][0][ks], c0);
                c1 = mfma16(piece[wt][1][ks], colB[wt][1][ks], c1);
            }
            Ct[wt][0] = c0; Ct[wt][1] = c1;
        }
    };

    // de-phase the waves (development knob p.exp: units of 512 cycles per wave index / workgroup parity)
    if ((p.exp >> 8) > 0) {
        const int nsl = ((wave & 3) + 4 * (int)(blockIdx.x & 1)) * (p.exp >> 8);
        for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(8);
    }
    if (wsub < nrt) { request(wsub); stage_b(wsub); reduce(); }

    for (int rt = wsub; rt < nrt; rt += NW) {
        float* tile = img + (size_t)rt * 16 * W;
        const int rows = min(16, H - 16 * rt);
        const int phase = (int)((reinterpret_cast<uintptr_t>(tile) >> 2) & 31);
        const int rowbase = phase + r16 * W;
        const bool row_ok = r16 < rows;
        const int nfast = rows == 16 ? max(1, min(nwt, (wfast_hi + 1) >> 4)) : 1;
        const bool more = rt + NW < nrt;
        // the NEXT tile's addend operands are requested now, straight behind the previous tile's stores: they land while this tile's
        // column loop runs (that is also the time those stores have to drain - loads and stores share one in-order counter)
        if (EARLY) request_rows(more ? rt + NW : rt);        // (last tile: valid addresses, results unused - no control flow around the loads)

        // ---- stage A' + stage 2' of the addend, column tile by column tile; tile wt - 1 is staged under tile wt's MFMAs.
        // FULL: all 16 rows exist - column tiles 1 .. NWTM - 2 need no guards, the last one by a uniform test
        asm volatile("" ::: "memory");          // the (tile-invariant) twiddle reads of the unrolled column loop stay inside the tile loop: registers
        auto columns = [&](auto full_t) {
            constexpr bool FULL = decltype(full_t)::value;
            f32x4 Ey[2], Dy[2], aL[2], aR[2];
            float2 tw[2][KS];
#pragma unroll
            for (int sp = 0; sp < KS; ++sp) tw[0][sp] = tabLane[sp * 64];
            auto stage = [&](int wt, bool fast, const f32x4& E, const f32x4& D, const f32x4& al, const f32x4& ar) {
                const int w0 = 16 * wt + 4 * kk;
                const f32x4 yl = E - D + al;
                const f32x4 yr = E + D + ar;
                if (fast) {                                                    // (uniform) interior tile of a full row tile
                    float* pl = buf + rowbase + w0;
                    float* pr = buf + rowbase + W - w0 - 3;
                    pl[0] = yl[0]; pl[1] = yl[1]; pl[2] = yl[2]; pl[3] = yl[3];
                    pr[3] = yr[0]; pr[2] = yr[1]; pr[1] = yr[2]; pr[0] = yr[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int w = w0 + e;
                        buf[(row_ok && w <= Wh) ? rowbase + w : dump] = yl[e];
                        buf[(row_ok && w >= 1 && w <= wfast_hi) ? rowbase + W - w : dump] = yr[e];
                    }
                }
            };
            auto is_fast = [&](int wt) { return FULL && wt >= 1 && (wt < NWTM - 1 || wt < nfast); };
#pragma unroll
            for (int wt = 0; wt < NWTM; ++wt) {
                const int cur = wt & 1, nx = cur ^ 1;
                const float2* nxt = tabLane + (size_t)(wt + 1 < NWTM ? wt + 1 : wt) * (KS * 64);
#pragma unroll
                for (int sp = 0; sp < KS; ++sp) tw[nx][sp] = nxt[sp * 64];
                __builtin_amdgcn_sched_barrier(0);
                f32x4 E = f32x4{0, 0, 0, 0}, D = f32x4{0, 0, 0, 0}, al = f32x4{0, 0, 0, 0}, ar = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int sp = 0; sp < (KS > ADD_KE ? KS : ADD_KE); ++sp) {
                    if (sp < KS) {
                        E = mfma16(tw[cur][sp].x, Ur[sp >> 2][sp & 3], E);
                        D = mfma16(tw[cur][sp].y, Ui[sp >> 2][sp & 3], D);
                    }
                    if (sp < ADD_KE && !(ko & 16)) {
                        al = mfma16(Ct[wt][0][sp], rowB[sp], al);
                        ar = mfma16(Ct[wt][1][sp], rowB[sp], ar);
                    }
                }
                Ey[cur] = E; Dy[cur] = D; aL[cur] = al; aR[cur] = ar;
                // the NEXT row tile's pieces of this column tile, spread over the column loop: the previous tile's stores are still
                // draining through the same path, two loads at a time slip in between them (all 28 at the top of the tile: 343 us)
                if (EARLY) request_piece(wt);
                if (wt >= 1) stage(wt - 1, is_fast(wt - 1), Ey[nx], Dy[nx], aL[nx], aR[nx]);
            }
            stage(NWTM - 1, is_fast(NWTM - 1), Ey[(NWTM - 1) & 1], Dy[(NWTM - 1) & 1], aL[(NWTM - 1) & 1], aR[(NWTM - 1) & 1]);
        };
        if (rows == 16) columns(std::true_type{}); else columns(std::false_type{});

        // ---- next tile: operands requested, column stage and stage 1' done BEFORE this tile's stores are issued (header)
        if (more) {
            if (!EARLY) request(rt + NW);
            stage_b(rt + NW);
            reduce();
        }

        // ---- the tile goes out as whole 128-byte lines: LDS index i <-> memory gbase[i]
        float* gbase = tile - phase;
        const int total = phase + rows * W;
        const bool first = !chain || rt == 0, last = !chain || !more;
        const int lo = first ? phase : 0;
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
        const int nfull = (ko & 4) ? 0 : (hi >> 8);
        if (!(ko & 4)) store_guarded(4 * lane);
        int it = 1;
        for (; it + 4 <= nfull; it += 4) {
            const float* src = buf + 256 * it + 4 * lane;
            float* dst = gbase + 256 * it + 4 * lane;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 256);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + 512);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + 768);
            *reinterpret_cast<f32x4*>(dst) = v0;
            *reinterpret_cast<f32x4*>(dst + 256) = v1;
            *reinterpret_cast<f32x4*>(dst + 512) = v2;
            *reinterpret_cast<f32x4*>(dst + 768) = v3;
        }
        for (; it < nfull; ++it)
            *reinterpret_cast<f32x4*>(gbase + 256 * it + 4 * lane) = *reinterpret_cast<const f32x4*>(buf + 256 * it + 4 * lane);
        if (nfull >= 1) store_guarded(256 * nfull + 4 * lane);
        if (!last) {
            const int rem = total & 31;
            float v = 0.f;
            if (lane < rem) v = buf[(total & ~31) + lane];
            if (lane < rem) buf[lane] = v;
        }
    }
}

// ---- launcher side: the form applies where K3-FT does (same LDS geometry), the column tiles fit the unrolled loop and the operand
// tables are given
static bool inv_add_shape_ok(const Dft2dParams& p) {
    const int nwt = ((p.W >> 1) + 16) >> 4;
    const int KS = (p.m2 + 3) / 4, KSK = (p.m1 + 4) >> 2;
    // (the unrolled column loop is compiled for 7 and 14 column tiles: rows of 192 .. 223 and 416 .. 447 elements)
    return (nwt == 7 || nwt == 14) && KS <= 6 && KSK >= 1 && KSK <= 7 && !p.bf16 && p.add_Ws >= 4 * ADD_KE && p.add_Hs >= 1;
}

template <int KS, int KSK, int NWTM>
static int launch_inv_add(Dft2dParams p, const InvGeometry& g, hipStream_t s) {
    auto k = dft2d_inv_ft_add_kernel<KS, KSK, NWTM>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), g.lds, lds_slot)) { set_error("dft2d_inv_add: cannot raise dynamic LDS to %zu", g.lds); return -4; }
    p.nw = g.nw;
    p.rev = next_sweep_reversed(SWEEP_K3);
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_inv_ft_add_kernel<%d, %d, %d>", KS, KSK, NWTM);
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * 4.0 + (double)p.add_Hs * p.add_Ws * 4.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3((p.n_img + g.g - 1) / g.g), dim3(64 * g.nw * g.g), g.lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_inv_add launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

template <int KS, int KSK>
static int launch_inv_add_t(const Dft2dParams& p, hipStream_t s) {
    InvGeometry ft;
    if (!inv_ft_geometry(p, KS, &ft)) { set_error("dft2d_inv_add: the full-tile form does not apply to %dx%d", p.H, p.W); return -3; }
    const int nwt = ((p.W >> 1) + 16) >> 4;
    if (nwt == 7) return launch_inv_add<KS, KSK, 7>(p, ft, s);
    return launch_inv_add<KS, KSK, 14>(p, ft, s);
}

}  // namespace uno
