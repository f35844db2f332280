// K1, half-tile form - pruned forward 2-D DFT, image tiles staged in LDS by direct-to-LDS loads, two waves per SIMD.
//
// What the measurements of this round say about K1:
//   * the register-path kernel (dft2d_fwd_kernel.h) reads its operand as 16 rows x 64-byte pieces at 4-byte alignment: every
//     instruction touches 16-32 cache lines and uses half of each, the wave has 6 KB in flight, and with its input cold in HBM
//     (which is how it runs inside a block) it takes 257 us for 726 MB = 2.8 TB/s;
//   * the full-tile form (dft2d_fwd_ft_kernel.h) fetches whole lines - a 16 x W tile is ONE contiguous run - but 27 KB of tile
//     per wave leave room for one wave per SIMD only, and a lone wave runs its LDS reads, its MFMAs and its tile wait back
//     to back: 268 us.
// This form keeps the whole-line fetches and gets the second wave per SIMD by holding only HALF a tile per wave: the row
// stage sums over column pairs (w, W - w), and the pairs split into an OUTER half (the first / last ~W/4 columns of every
// row) and an INNER half (the middle ~W/2 columns).  In memory the outer half of a 16-row tile is 15 contiguous runs "end of
// row r-1 | start of row r" (+ two half runs), the inner half 16 runs "middle of row r", each ~0.85 KB: one
// buffer_load_dwordx4 ... lds per run.  A wave alternates
//     wait(outer) - row stage over the outer pairs - request(inner) - wait - row stage over the inner pairs - request(next
//     tile's outer) - column stage - ...
// with ONE 14 KB buffer, so both waits are exposed to the wave - and hidden by the other wave of the SIMD, which is in its
// compute phases then.  Twiddles come from the operand-layout table in LDS shared by the workgroup's 8 waves.
#pragma once
#include "uno_common.h"
#include "dft2d_fwd_ft_kernel.h"
#include <algorithm>
#include <cstdio>

namespace uno {

constexpr size_t HT_LDS_LIMIT = 160 * 1024;         // a workgroup may own the whole LDS of the CU
constexpr int HT_WAVES = 8;
constexpr int HT_AUX = 2;               // cache policy of the tile loads: 2 = non-temporal (each line is read once; measured 240 -> 193 us at the C2 block)

struct HtSplit { int cs, QL, QR, len_in, seg, s0, off0; };
// outer half = column 0 + the pairs of chunks [0, cs): left columns [0, QL), right columns [W - QR, W); inner = the rest.
// LDS slots of one wave: slot 0 (s0 floats) holds run 16 (right part of row 15) at its head and run 0 (left part of row 0) from
// off0 on; slot k >= 1 (seg floats, at s0 + (k - 1) seg) holds run k = [right part of row k-1 | left part of row k].  The inner
// half reuses the slots, one row each.  seg = 4 (mod 8) floats keeps the 16 rows of an operand read off each other's banks.
__host__ __device__ inline HtSplit ht_split(int W) {
    const int P = (W - 1) >> 1, nfull = P >> 4;
    HtSplit s;
    int cs = (W - 2 + 32) / 64;
    cs = cs < 1 ? 1 : cs;
    s.cs = cs > nfull ? nfull : cs;
    s.QR = 16 * s.cs;
    s.QL = s.QR + 1;
    s.len_in = W - s.QL - s.QR;
    s.off0 = (s.QR + 3 + 3) & ~3;
    const int a = s.QL + s.QR + 3, b = s.len_in + 3;
    s.seg = ((a > b ? a : b) + 3) & ~3;
    if ((s.seg & 7) == 0) s.seg += 4;
    const int c = s.off0 + 3 + s.QL;
    s.s0 = ((c > s.seg ? c : s.seg) + 3) & ~3;
    return s;
}

template <int NT, int MT, int R4>
__global__ __launch_bounds__(64 * HT_WAVES) void dft2d_fwd_ht_kernel(Dft2dParams p) {
    constexpr int NTF = R4 > 0 ? NT - 1 : NT;       // full 16-mode streams
    constexpr int NQ = R4 > 0 ? R4 : 1;
    constexpr int NTFA = NTF > 0 ? NTF : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int NWT = nthreads >> 6;
    const int NW = p.nw;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const unsigned H8 = 8u * H;

    const int P = (W - 1) >> 1;
    const int nfull = P >> 4;
    const int prem = P - (nfull << 4);
    const int ntail = prem + 1 + ((W & 1) ? 0 : 1);
    const int tailsteps = (ntail + 3) >> 2;
    const int nk = 4 * nfull + tailsteps;
    const int nka = 4 * nfull + FT_TAILMAX;
    const HtSplit sp = ht_split(W);
    const int cs = sp.cs, QL = sp.QL, QR = sp.QR, LEN_IN = sp.len_in, SEG = sp.seg, S0 = sp.s0, OFF0 = sp.off0;
    auto slot_base = [&](int k) { return k == 0 ? 0 : S0 + (k - 1) * SEG; };

    const int buf_stride = S0 + 15 * SEG;
    float* sBuf = reinterpret_cast<float*>(smem);                              // [NWT][S0 + 15 SEG]
    float2* sTabF = reinterpret_cast<float2*>(sBuf + (size_t)NWT * buf_stride);     // [nka][NTF][64]
    float2* sTab4 = sTabF + (size_t)nka * NTF * 64;                            // [nka][R4][16]
    float2* sTwH = sTab4 + (size_t)nka * R4 * 16;

    const int slot = wave / NW, wsub = wave - slot * NW;
    const int image = sweep_x(p.rev) * (NWT / NW) + slot;
    const bool active = image < p.n_img;
    const int nrt = (H + 15) >> 4;
    float* buf = sBuf + (size_t)wave * buf_stride;

    // buffer resource = [128-byte aligned start of the image, end of the tensor): offsets are non-negative, reads past the
    // tensor return zero
    const float* timg = p.in + (size_t)(active ? image : 0) * H * W;
    const uintptr_t ibase = reinterpret_cast<uintptr_t>(timg) & ~uintptr_t(127);
    const int a0 = (int)((reinterpret_cast<uintptr_t>(timg) - ibase) >> 2);
    const unsigned long long span = reinterpret_cast<uintptr_t>(p.in + (size_t)p.n_img * H * W) - ibase;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(ibase), 0, (int)(unsigned)std::min<unsigned long long>(span, 0xffffffffull), 0x00020000);
    // one run: floats [m, m + len) from the aligned base -> LDS floats [dst + (m & 3), ...): the fetch starts at the 16-byte
    // boundary below m, so element e of the run lands at dst + (m & 3) + e
    auto fetch_run = [&](int m, int len, float* dst) {
        const int ph = m & 3;
        if (4 * lane < ph + len)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, (unsigned)((m - ph + 4 * lane) * 4), 0, 0, HT_AUX);
    };
    auto request_outer = [&](int rt) {
        const int toff = a0 + rt * 16 * W;
        const int rows = min(16, H - 16 * rt);
        fetch_run(toff, QL, buf + OFF0);                                        // run 0: left part of row 0
        for (int k = 1; k < rows; ++k) fetch_run(toff + k * W - QR, QR + QL, buf + slot_base(k));      // right of row k-1 | left of row k
        // right part of the last valid row: run `rows` (slot rows, or slot 0 for a full tile)
        fetch_run(toff + rows * W - QR, QR, rows == 16 ? buf : buf + slot_base(rows));
    };
    auto request_inner = [&](int rt) {
        const int toff = a0 + rt * 16 * W;
        const int rows = min(16, H - 16 * rt);
        for (int k = 0; k < rows; ++k) fetch_run(toff + k * W + QL, LEN_IN, buf + slot_base(k));
    };
    // rows past the image (last tile only): their LDS slots are zero-filled - the column stage multiplies them by a zero twiddle,
    // and 0 * (stale LDS bits) must not be NaN
    auto zero_invalid = [&](int toff, int rows, bool outer) {
        // outer layout: slot `rows` keeps the valid right part of row rows-1 at its head (phase + QR floats)
        const int keep_first = outer ? ((toff + rows * W - QR) & 3) + QR : 0;
        for (int k = rows; k < 16; ++k) {
            const int n = k == 0 ? S0 : SEG;
            for (int i = (k == rows ? keep_first : 0) + lane; i < n; i += 64) buf[slot_base(k) + i] = 0.f;
        }
        if (outer)                          // slot 0's head (row 15's right part in a full tile) is the invalid row 15 here
            for (int i = lane; i < OFF0; i += 64) buf[i] = 0.f;
    };
    if (active && wsub < nrt) request_outer(wsub);

    // ---- tables (built while the first half tile is on its way)
    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];
    for (int e = tid; e < nk * 64; e += nthreads) {
        const int ln = e & 63, q = e >> 6, ks = ln >> 4;
        unsigned w;
        if (q < 4 * nfull) {
            w = 1u + 16u * (q >> 2) + 4u * ks + (q & 3);
        } else {
            const int qt = 4 * (q - 4 * nfull) + ks;
            w = qt < prem ? 1u + 16u * nfull + qt : ((qt == prem + 1 && !(W & 1)) ? (unsigned)(W >> 1) : 0u);
        }
#pragma unroll
        for (int t = 0; t < NTF; ++t) {
            const unsigned l = (unsigned)min(16 * t + (ln & 15), m2 - 1);
            sTabF[((size_t)q * NTF + t) * 64 + ln] = p.twW[(w * l) % (unsigned)W];
        }
        if ((ln & 12) == 0) {
#pragma unroll
            for (int g = 0; g < R4; ++g) {
                const unsigned l = (unsigned)min(16 * NTF + 4 * g + (ln & 3), m2 - 1);
                sTab4[((size_t)q * R4 + g) * 16 + 4 * ks + (ln & 3)] = p.twW[(w * l) % (unsigned)W];
            }
        }
    }
    __syncthreads();

    // Column stage in +-k PAIRED form where that saves row tiles (MT >= 3): the kept rows are the frequencies +k (lo corner row k,
    // k < m1) and -k (hi corner row 2 m1 - k, 1 <= k <= m1), and with C_k = sum_h cos(theta_k h) T[h], S_k = sum_h sin(theta_k h) T[h]
    //     X[+k] = C_k - i S_k,      X[-k] = C_k + i S_k
    // so the stage runs over m1 + 1 values of k (MP tiles of 16) with REAL twiddles instead of 2 m1 rows (MT tiles) with complex
    // ones: 4 MP instead of 4 MT MFMAs per (k-step, mode tile) - 8 against 12 at modes1 = 18 / 20 (the column stage is ~40 % of
    // this kernel's MFMA cycles), 12 against 16 at modes1 = 32.  K3 has used the same pairing since round 1.
    constexpr bool PAIR = MT >= 3 && NT * MT < 12;      // (NT, MT) = (3, 4), (3, 5): the four accumulator sets no longer fit 256 VGPRs
    constexpr int MP = PAIR ? MT / 2 + 1 : MT;          // >= ceil((m1 + 1) / 16) for every m1 <= 8 MT
    int Kj[MP];
    bool jvalid[MP];
#pragma unroll
    for (int mt = 0; mt < MP; ++mt) {
        const int j = 16 * mt + r16;
        if constexpr (PAIR) {
            jvalid[mt] = j <= m1;                       // k = j: 0 .. m1
            Kj[mt] = jvalid[mt] ? j : 0;
        } else {
            jvalid[mt] = j < 2 * m1;
            Kj[mt] = jvalid[mt] ? corner_freq(j, m1, H) : 0;
        }
    }
    // !PAIR: Xr / Xi = Re / Im of the spectrum rows, Yr / Yi unused.  PAIR: Xr = Re C, Xi = Re S, Yr = -Im C, Yi = -Im S
    // (the row stage hands over Tn = -Im T)
    f32x4 Xr[MP][NT], Xi[MP][NT], Yr[PAIR ? MP : 1][NT], Yi[PAIR ? MP : 1][NT];
#pragma unroll
    for (int mt = 0; mt < MP; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Xr[mt][t] = f32x4{0, 0, 0, 0}; Xi[mt][t] = f32x4{0, 0, 0, 0};
            if (PAIR || mt == 0) { Yr[PAIR ? mt : 0][t] = f32x4{0, 0, 0, 0}; Yi[PAIR ? mt : 0][t] = f32x4{0, 0, 0, 0}; }
        }

    const float2* tabF = sTabF + lane;
    const float2* tab4 = sTab4 + 4 * kk + (lane & 3);

    if (active) {
        for (int rt = wsub; rt < nrt; rt += NW) {
            const int toff = a0 + rt * 16 * W;
            const int rows = min(16, H - 16 * rt);
            // where this lane's row lives in the two layouts (phases: fetch_run)
            const int ph0 = toff & 3;
            const int phl = (toff + r16 * W - QR) & 3;                  // slot r16 (r16 >= 1): [right of row r16-1 | left of row r16]
            const int phr = (toff + (r16 + 1) * W - QR) & 3;            // run r16 + 1 holds the right part of row r16
            const float* lptr = buf + (r16 == 0 ? OFF0 + ph0 : slot_base(r16) + phl + QR);      // column c of row r16, c < QL
            const int rslot = (r16 + 1 == rows && rows < 16) ? rows : ((r16 + 1) & 15);     // full tile: row 15's right part is in slot 0
            const float* rptr = buf + slot_base(rslot) + phr;                // column W - QR + j of row r16, j < QR
            const int phi = (toff + r16 * W + QL) & 3;
            const float* iptr = buf + slot_base(r16) + phi;                  // column QL + e of row r16, e < LEN_IN

            f32x4 Tr[NT], Tn[NT];           // Tn = -Im T
            f32x4 Qr[NQ], Qn[NQ];           // 4x4x1 accumulators of the 4-mode groups (R4 > 0)
#pragma unroll
            for (int t = 0; t < NT; ++t) { Tr[t] = f32x4{0, 0, 0, 0}; Tn[t] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int g = 0; g < NQ; ++g) { Qr[g] = f32x4{0, 0, 0, 0}; Qn[g] = f32x4{0, 0, 0, 0}; }
#define UNO_HT_MFMA(E_, D_, TWF_, TW4_)                                                   \
    do {                                                                                  \
        _Pragma("unroll") for (int t = 0; t < NTF; ++t) {                                 \
            Tr[t] = mfma16((E_), (TWF_)[t].x, Tr[t]);                                     \
            Tn[t] = mfma16((D_), (TWF_)[t].y, Tn[t]);                                     \
        }                                                                                 \
        _Pragma("unroll") for (int g = 0; g < R4; ++g) {                                  \
            Qr[g] = ft_mfma4((E_), (TW4_)[g].x, Qr[g]);                                   \
            Qn[g] = ft_mfma4((D_), (TW4_)[g].y, Qn[g]);                                   \
        }                                                                                 \
    } while (0)
            // chunks [c_lo, c_hi) of the row stage: lane (row r16, k-slot kk) owns column pairs w = 1 + 16 c + 4 kk + s; pl / pr
            // point at this lane's four left / mirrored columns of chunk c_lo and move by +-16 floats per chunk
            // (two operand sets in ping-pong - round 5: the rolled loop copied the next chunk's 8 image values and 8 NT twiddle
            // registers into the current set after every MFMA block, ~24 v_mov per 8 + 8 MFMAs that no MFMA covered: a wave's
            // VALU instructions run under its OWN MFMAs only when they sit between them in program order)
            struct RowOps { float xl[4], xr[4]; float2 twF[4][NTFA], tw4[4][NQ]; };
            auto row_stage = [&](const float* pl, const float* pr, int c_lo, int c_hi) {
                RowOps A, B;
                auto load_ops = [&](RowOps& o, int c) {                    // operands of chunk c (clamped to the last chunk of the range)
                    const int dn = 16 * (min(c, c_hi - 1) - c_lo);
                    const float2* tf = tabF + (size_t)(4 * c) * (NTF * 64);
                    const float2* t4 = tab4 + (size_t)(4 * c) * (R4 * 16);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        o.xl[s] = pl[dn + s]; o.xr[s] = pr[-dn + s];
#pragma unroll
                        for (int t = 0; t < NTF; ++t) o.twF[s][t] = tf[(s * NTF + t) * 64];
#pragma unroll
                        for (int g = 0; g < R4; ++g) o.tw4[s][g] = t4[(s * R4 + g) * 16];
                    }
                };
                auto chunk = [&](RowOps& cur, RowOps& nxt, int c) {
                    load_ops(nxt, c + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float E = cur.xl[s] + cur.xr[3 - s];
                        const float D = cur.xl[s] - cur.xr[3 - s];
                        UNO_HT_MFMA(E, D, cur.twF[s], cur.tw4[s]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                load_ops(A, c_lo);
                int c = c_lo;
                for (; c + 2 <= c_hi; c += 2) { chunk(A, B, c); chunk(B, A, c + 1); }
                if (c < c_hi) chunk(A, B, c);
            };

            // ---- outer half
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (rows < 16) zero_invalid(toff, rows, true);
            const float x0 = lptr[0];                                   // column 0: a tail element, kept in a register
            row_stage(lptr + 1 + 4 * kk, rptr + QR - 4 - 4 * kk, 0, cs);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            request_inner(rt);
            // ---- inner half
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (rows < 16) zero_invalid(toff, rows, false);
            if (cs < nfull) row_stage(iptr + 4 * kk, iptr + LEN_IN - 4 - 4 * kk, cs, nfull);
            // tail k-steps: pairs beyond the last full chunk (inner columns), w = 0 (x0), the Nyquist column (inner)
            {
                const float2* tf = tabF + (size_t)(4 * nfull) * (NTF * 64);
                const float2* t4 = tab4 + (size_t)(4 * nfull) * (R4 * 16);
#pragma unroll
                for (int s = 0; s < FT_TAILMAX; ++s) {
                    if (s < tailsteps) {
                        const int q = 4 * s + kk;
                        const bool pair = q < prem;
                        const bool nyq = (q == prem + 1) && !(W & 1);
                        const int wl = pair ? 1 + 16 * nfull + q : (W >> 1);       // inner column of the left element (pair / Nyquist)
                        const float vin = iptr[min(max(wl - QL, 0), LEN_IN - 1)];
                        const float vr = iptr[min(max(W - wl - QL, 0), LEN_IN - 1)];
                        const float TL = (pair || nyq) ? vin : (q == prem ? x0 : 0.f);
                        const float TR = pair ? vr : 0.f;
                        float2 twF[NTFA], tw4[NQ];
#pragma unroll
                        for (int t = 0; t < NTF; ++t) twF[t] = tf[(s * NTF + t) * 64];
#pragma unroll
                        for (int g = 0; g < R4; ++g) tw4[g] = t4[(s * R4 + g) * 16];
                        const float E = TL + TR;
                        const float D = TL - TR;
                        UNO_HT_MFMA(E, D, twF, tw4);
                    }
                }
            }
#undef UNO_HT_MFMA
            // the row stage is done with the buffer: request the next tile's outer half, it lands during the column stage
            if (rt + NW < nrt) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                request_outer(rt + NW);
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

            // ---- stage B: X[j][l] += exp(-i theta(j,h)) * T[h][l], h = 16 rt + 4 kk + s
            unsigned idxB[MP];
            float2 twB[MP];
#pragma unroll
            for (int mt = 0; mt < MP; ++mt) {
                const unsigned i0 = 8u * (((unsigned)Kj[mt] * (unsigned)(16 * rt + 4 * kk)) % (unsigned)H);
                twB[mt] = lds_tw(sTwH, i0);
                idxB[mt] = wrap_add(i0, 8u * (unsigned)Kj[mt], H8);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool hvalid = (16 * rt + 4 * kk + s) < H;
                float2 twBn[MP];
#pragma unroll
                for (int mt = 0; mt < MP; ++mt) {
                    twBn[mt] = lds_tw(sTwH, idxB[mt]);
                    idxB[mt] = wrap_add(idxB[mt], 8u * (unsigned)Kj[mt], H8);
                }
#pragma unroll
                for (int mt = 0; mt < MP; ++mt) {
                    const bool v = hvalid && jvalid[mt];
                    const float ac = v ? twB[mt].x : 0.f;
                    if constexpr (PAIR) {
                        const float as = v ? twB[mt].y : 0.f;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            Xr[mt][t] = mfma16(ac, Tr[t][s], Xr[mt][t]);         // Re C
                            Yr[mt][t] = mfma16(ac, Tn[t][s], Yr[mt][t]);         // -Im C
                            Xi[mt][t] = mfma16(as, Tr[t][s], Xi[mt][t]);         // Re S
                            Yi[mt][t] = mfma16(as, Tn[t][s], Yi[mt][t]);         // -Im S
                        }
                    } else {
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
                }
#pragma unroll
                for (int mt = 0; mt < MP; ++mt) twB[mt] = twBn[mt];
            }
        }
    }

    // ---- several waves per image: deterministic tree reduction of the partial spectra through the (now free) buffers
    if (NW > 1) __syncthreads();            // every wave of the workgroup is done with its buffer
    // one pass per accumulator pair ((Xr, Xi), and in the paired form (Yr, Yi)): MP * NT * 8 * 64 floats fit a wave's buffer
    auto reduce_pair = [&](f32x4 (*A)[NT], f32x4 (*Bv)[NT], int stride) {
        if (wsub >= stride && wsub < 2 * stride) {
            float* dst = sBuf + (size_t)(wave - stride) * buf_stride;         // the partner's buffer
#pragma unroll
            for (int mt = 0; mt < MP; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[((mt * NT + t) * 8 + r) * 64 + lane] = A[mt][t][r];
                        dst[((mt * NT + t) * 8 + 4 + r) * 64 + lane] = Bv[mt][t][r];
                    }
        }
        __syncthreads();
        if (wsub < stride && wsub + stride < NW) {
#pragma unroll
            for (int mt = 0; mt < MP; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        A[mt][t][r] += buf[((mt * NT + t) * 8 + r) * 64 + lane];
                        Bv[mt][t][r] += buf[((mt * NT + t) * 8 + 4 + r) * 64 + lane];
                    }
        }
        __syncthreads();
    };
    for (int stride = 2; stride >= 1; stride >>= 1) {
        if (stride >= NW) continue;
        reduce_pair(Xr, Xi, stride);
        if constexpr (PAIR) reduce_pair(Yr, Yi, stride);
    }

    if (active && wsub == 0) {
        float2* out = reinterpret_cast<float2*>(p.out) + spectrum_index(p, image) * 2 * m1 * m2;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int l = 16 * t + r16;
            if (l >= m2) continue;
            const float cs_ = p.scale * (p.herm ? herm_weight(l, W) : 1.0f);
#pragma unroll
            for (int mt = 0; mt < MP; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * mt + 4 * kk + r;
                    if constexpr (PAIR) {
                        // k = j:  X[+k] = C - i S = (Re C - (-Im S)... with Yr = -Im C, Yi = -Im S:  Re = Xr - Yi, Im = -Yr - Xi;  X[-k]: S -> -S
                        const float cr = Xr[mt][t][r], sr = Xi[mt][t][r], cn = Yr[PAIR ? mt : 0][t][r], sn = Yi[PAIR ? mt : 0][t][r];
                        if (j < m1) {
                            const float f = (p.mask && !row_survives(j, m1, H)) ? 0.f : cs_;
                            out[(size_t)j * m2 + l] = make_float2((cr - sn) * f, (-cn - sr) * f);
                        }
                        if (j >= 1 && j <= m1) {
                            const int jm = 2 * m1 - j;
                            const float f = (p.mask && !row_survives(jm, m1, H)) ? 0.f : cs_;
                            out[(size_t)jm * m2 + l] = make_float2((cr + sn) * f, (-cn + sr) * f);
                        }
                    } else {
                        if (j < 2 * m1) {
                            const float f = (p.mask && !row_survives(j, m1, H)) ? 0.f : cs_;
                            out[(size_t)j * m2 + l] = make_float2(Xr[mt][t][r] * f, Xi[mt][t][r] * f);
                        }
                    }
                }
        }
    }
}

// ---- launcher side
static size_t fwd_ht_lds_bytes(const Dft2dParams& p, int NTF, int R4, int waves) {
    const int P = (p.W - 1) >> 1, nfull = P >> 4;
    const size_t nka = (size_t)4 * nfull + FT_TAILMAX;
    const HtSplit sp = ht_split(p.W);
    return (size_t)waves * (sp.s0 + 15 * sp.seg) * 4 + nka * ((size_t)NTF * 512 + (size_t)R4 * 128) + (size_t)p.H * 8;
}

// NW in {1, 2, 4} waves per image, G = 8 / NW images per workgroup (fewer when the images do not fill the CUs)
static bool fwd_ht_geometry(const Dft2dParams& p, int NT, int MT, int R4, FwdFtGeometry* out) {
    const int NTF = R4 > 0 ? NT - 1 : NT;
    const int P = (p.W - 1) >> 1, nfull = P >> 4;
    if (p.bf16 || p.rowfreq || nfull < 2 || p.W <= FT_MAXW) return false;
    // measured (1024 images, in-block, input cold): 421^2 227 us against 257 for the register path, 446^2 218 / 230, 223^2 51 / 47:
    // short rows make short runs (a run of the inner half of a 223-wide row is 0.4 KB), the register path keeps those
    // round 3: from 200 columns.  The 51 / 47 us above were taken with inputs that a standalone loop keeps partly cache-resident;
    // inside a training step (inputs cold in HBM, tools/dev/steplaunches.py) the 223^2 layers of the Darcy model take 126-131 us here
    // against 150-162 us on the register path at modes 18 (with the paired column stage) and 113-115 against 128-160 us at modes 8
    if (p.W < 200) return false;
    const HtSplit sp = ht_split(p.W);
    if (sp.QL + sp.QR + 3 > 256 || sp.len_in + 3 > 256 || sp.len_in < 8) return false;      // a run is one 64-lane x 16-byte fetch
    const int MPh = (MT >= 3 && NT * MT < 12) ? MT / 2 + 1 : MT;                          // row tiles of the (paired) column stage
    if ((size_t)MPh * NT * 8 * 64 > (size_t)(sp.s0 + 15 * sp.seg)) return false;         // reduction slots must fit a buffer
    const int nrt = (p.H + 15) / 16, cus = ft_device_cu_count();
    long long best_cost = -1;
    for (int nw = 1; nw <= 4 && nw <= nrt; nw *= 2) {
        int g = HT_WAVES / nw;
        while (g > 1 && (long long)(p.n_img + g - 1) / g < cus) --g;
        while (g > 1 && fwd_ht_lds_bytes(p, NTF, R4, nw * g) > HT_LDS_LIMIT) --g;
        const size_t lds = fwd_ht_lds_bytes(p, NTF, R4, nw * g);
        if (lds > HT_LDS_LIMIT) continue;
        const long long per_cu = std::max<long long>(1, std::min<long long>((long long)(HT_LDS_LIMIT / lds), 16 / (nw * g)));
        if (per_cu * nw * g < 6 && (long long)p.n_img * nw >= 8LL * cus) continue;      // the point of this form is two waves per SIMD
        const long long groups = (p.n_img + g - 1) / g;
        const long long rounds = (groups + cus * per_cu - 1) / (cus * per_cu);
        const long long cost = rounds * ((nrt + nw - 1) / nw);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; *out = FwdFtGeometry{nw, g, lds}; }
    }
    return best_cost >= 0;
}

template <int NT, int MT, int R4>
static int launch_fwd_ht(Dft2dParams p, const FwdFtGeometry& g, hipStream_t s) {
    auto k = dft2d_fwd_ht_kernel<NT, MT, R4>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), g.lds, lds_slot)) { set_error("dft2d_fwd: cannot raise dynamic LDS to %zu", g.lds); return -4; }
    p.nw = g.nw;
    p.rev = next_sweep_reversed(SWEEP_K1);
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_fwd_ht_kernel<%d, %d, %d>", NT, MT, R4);
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * 4.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3((p.n_img + g.g - 1) / g.g), dim3(64 * g.nw * g.g), g.lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_fwd launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
