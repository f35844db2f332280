// K1v / K3v - the three pruned transforms of the 3-D layer (SpectralConv3d_Uno.forward, reference integral_operators.py:395-427:
// rfftn over (dim1, dim2, dim3) restricted to the four corners, and irfftn of the zero-padded corners) with ONE workgroup per
// (sample, channel) VOLUME, instead of the plane-batched K1p / K3p + the leading-axis K5 / K6 with a truncated per-plane
// spectrum (n_vol x D1 x 2 m2 x m3 complex: 33 MB + 17 MB of intermediates on the 235 MB block of config C4) between them.
//
// The per-plane spectra of a volume (D1 planes x 2 m2 x m3 complex: 128 KB at 64 x 32 x 8) live in LDS; the workgroup's 16 waves
// deal the planes among themselves, transform them (T axis, then dim2) straight from global memory in MFMA operand layout
// (no LDS copy of the image: a lane's 16-byte piece of a row IS the A operand of four k-steps once the k-steps are numbered
// "column 4 g + e" instead of "column 4 e + g"), meet at one barrier and transform the leading axis out of LDS.
//
// Both complex axes (dim1 and dim2: N points, the 2 m corner rows k = -m .. m-1 kept) use the HALF-SHIFTED PAIRED form:
// with kappa = k + 1/2 the kept rows are +-kappa, kappa = 1/2 .. m - 1/2, and with y[h] = x[h] e^{+i pi h / N}
//
//     X[k] = sum_h y[h] e^{-2 pi i kappa h / N},        e^{-2 pi i kappa (N - h) / N} = -e^{+2 pi i kappa h / N}
//     X[+-kappa] = sum_{i < N/2} cos(theta_i) D'[i]  -+  i sum_{i < N/2} sin(theta_i) E'[i]          theta_i = 2 pi kappa i / N
//     D'[i] = y[i] - y[N - i] = c (x[i] + x[N-i]) + i s (x[i] - x[N-i]),   E'[i] = y[i] + y[N - i] = c (x[i] - x[N-i]) + i s (x[i] + x[N-i])
//     (c + i s = e^{i pi i / N});   i = 0:  D'[0] = x[0],  E'[0] = i x[N/2] with "sin" weight (-1)^m  (the slot sin(0) leaves free)
//
// i.e. two real-weight GEMMs with K = N/2 and M = m instead of four with K = N and M = 2 m: a quarter of the MFMA work of the plain
// form, for ~4 VALU operations per element (the e^{i pi h / N} twist and the sums / differences).  The inverse runs the same
// identities backwards.  Results equal K1p / K3p + K5 / K6 to f32 rounding (different summation order).
#include "uno_common.h"
#include <algorithm>
#include <cstdio>

namespace uno {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int VOL_WAVES = 16;
constexpr size_t VOL_LDS_LIMIT = 160 * 1024;
constexpr int VOL_MIN_VOLUMES = 48;     // one workgroup per volume; measured (tools/dev/vol3dtime.py, widths 8 and 16): still ahead of the plane path at 64 volumes

__device__ __forceinline__ float vol_xor1(float v) {       // the value held by lane ^ 1 (DPP quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

struct VolShape {
    int nslot1, nslot2;      // pair slots along dim1 / dim2: slot 0 = (0, N/2), slot i = (i, N - i)
    int U;                   // 16-slot tile pairs of a plane
    int MT2, MT1;            // 16-row tiles of kappa along dim2 / dim1
    int NBW, NARROW;         // dim3 in 16-column blocks: NBW with a 16-byte piece per lane (4 k-steps), + one 4-byte block (1 k-step)
    int KA;                  // k-steps of the T-axis stage
    int C2;                  // floats of a per-plane truncated spectrum: 2 m2 rows x m3 complex
    int RP;                  // its row pitch in LDS (floats), = 32 mod 64: the two k-slots of a half-wave's 8-byte B-operand reads fall on distinct banks
    int NT;                  // 32-column (16 complex) blocks of C2
    int nks1;                // k-steps of the leading-axis stage
    size_t lds;
};

static VolShape vol_shape(int D1, int D2, int D3, int m1, int m2, int m3) {
    VolShape g;
    g.nslot1 = (D1 + 1) / 2;
    g.nslot2 = (D2 + 1) / 2;
    g.U = (g.nslot2 + 15) / 16;
    g.MT2 = (m2 + 15) / 16;
    g.MT1 = (m1 + 15) / 16;
    const int nblk = (D3 + 15) / 16, rem = D3 - 16 * (nblk - 1);
    g.NARROW = rem <= 4 ? 1 : 0;
    g.NBW = nblk - g.NARROW;
    g.KA = 4 * g.NBW + g.NARROW;
    g.C2 = 2 * m2 * 2 * m3;
    g.NT = (g.C2 + 31) / 32;
    g.RP = 32 * g.NT;
    while ((g.RP & 63) != 32) g.RP += 32;
    g.nks1 = (g.nslot1 + 3) / 4;
    g.lds = (size_t)2 * g.nslot1 * g.RP * 4                      // sD, sE
            + (size_t)g.KA * 64 * 4                              // T-axis twiddles (B operand)
            + (size_t)g.U * 4 * g.MT2 * 64 * 8                   // dim2 (cos, sin) (A operand)
            + (size_t)g.U * 16 * 8                               // dim2 twist
            + (size_t)g.nks1 * g.MT1 * 64 * 8;                   // dim1 (cos, sin) (A operand)
    return g;
}

bool vol3d_fwd_applies(int n_vol, int D1, int D2, int D3, int m1, int m2, int m3) {
    if (n_vol < VOL_MIN_VOLUMES) return false;
    if (D1 < 4 || D2 < 4 || D3 < 2 || D1 > 128 || D2 > 64 || D3 > 32) return false;
    if (2 * m1 > D1 || 2 * m2 > D2 || m1 > 32 || m2 > 32 || 2 * m3 > 16 || m3 > D3 / 2 + 1) return false;      // no corner overlap
    const VolShape g = vol_shape(D1, D2, D3, m1, m2, m3);
    if (g.NBW < 1 || g.NBW > 2 || g.U > 2) return false;
    if ((long long)D1 * D2 * D3 * 4 >= (1LL << 31)) return false;
    return g.lds <= VOL_LDS_LIMIT;
}

// ---------------------------------------------------------------------------------------------------------------- K1v
// One plane of the volume -> its truncated spectrum in registers: Xp[mt][r] / Xm[mt][r] = rows +kappa / -kappa with
// kappa index 16 mt + 4 r + g (g = lane >> 4), column n = lane & 15 (re / im of T-mode n >> 1 interleaved).
template <int MT2, int NBW, bool NARROW>
__global__ __launch_bounds__(64 * VOL_WAVES) void dft3d_fwd_volume_kernel(Vol3dParams p, VolShape g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D1 = p.D1, D2 = p.D2, D3 = p.D3, m1 = p.m1, m2 = p.m2, m3 = p.m3;
    float* sD = reinterpret_cast<float*>(smem);                              // [nslot1][RP]
    float* sE = sD + (size_t)g.nslot1 * g.RP;
    float* sTwA = sE + (size_t)g.nslot1 * g.RP;                              // [KA][64]
    float2* sTwB = reinterpret_cast<float2*>(sTwA + g.KA * 64);              // [U * 4][MT2][64]
    float2* sTwist2 = sTwB + g.U * 4 * MT2 * 64;                             // [U * 16]
    float2* sTw1 = sTwist2 + g.U * 16;                                       // [nks1][MT1][64]
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = 64 * VOL_WAVES;

    // ---- phase 1: planes -> paired, twisted per-plane spectra in LDS
    const int vol = blockIdx.x;
    const size_t vol_elems = (size_t)D1 * D2 * D3;
    const float* vbase = p.in + (size_t)vol * vol_elems;
    const size_t after = ((size_t)(p.n_vol - vol)) * vol_elems * 4;          // bytes from this volume to the end of the tensor
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)std::min<size_t>(after, 0x7fffffffu), 0x00020000);
    const float sg = (lane & 1) ? 1.f : -1.f;
    const bool even2 = !(D2 & 1), even1 = !(D1 & 1);
    const int U = g.U;

    // rows of tile pair u: P = slot 16 u + n16, Q = its partner D2 - slot (slot 0: D2 / 2); rows past the last slot: clamped, their
    // twiddles are zero.  Byte offsets of this lane's pieces inside a plane:
    auto row_off = [&](int u, bool q) -> unsigned {
        int i = min(16 * u + n16, g.nslot2 - 1);
        int h = q ? (i == 0 ? D2 / 2 : D2 - i) : i;
        return (unsigned)(h * D3 + 4 * gq) * 4u;
    };
    unsigned offP[2], offQ[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { offP[u] = row_off(min(u, U - 1), false); offQ[u] = row_off(min(u, U - 1), true); }

    struct Piece { u32x4v w[NBW]; unsigned nrw; };
    Piece nP, nQ, cP, cQ;
    auto issue = [&](int plane, int u) {
        const unsigned sbase = (unsigned)plane * (unsigned)(D2 * D3) * 4u;
        const unsigned oP = u ? offP[1] : offP[0], oQ = u ? offQ[1] : offQ[0];
#pragma unroll
        for (int c = 0; c < NBW; ++c) {
            nP.w[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, oP + 64u * c, sbase, 0);
            nQ.w[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, oQ + 64u * c, sbase, 0);
        }
        if (NARROW) {
            // column 16 NBW + g: this lane's piece starts at column 4 g -> + (16 NBW - 3 g) columns
            nP.nrw = __builtin_amdgcn_raw_buffer_load_b32(rsrc, oP + 4u * (16 * NBW - 3 * gq), sbase, 0);
            nQ.nrw = __builtin_amdgcn_raw_buffer_load_b32(rsrc, oQ + 4u * (16 * NBW - 3 * gq), sbase, 0);
        }
    };

    // this wave's slots: wave, wave + 16, ...; per slot two planes (a = slot, b = partner), per plane U tile pairs
    const int my_slots = (g.nslot1 - wave + VOL_WAVES - 1) / VOL_WAVES;      // may be <= 0
    const int n_units = max(my_slots, 0) * 2 * U;
    auto unit_plane = [&](int q) -> int {
        const int slot = wave + VOL_WAVES * (q / (2 * U));
        const int second = (q / U) & 1;
        if (!second) return slot;
        return slot == 0 ? D1 / 2 : D1 - slot;                              // (odd D1, slot 0: plane D1 / 2 is loaded and ignored)
    };
    if (n_units > 0) issue(unit_plane(0), 0);            // the first rows are on their way while the tables are built

    // ---- tables
    for (int e = tid; e < g.KA * 64; e += nthreads) {
        const int ln = e & 63, ks = e >> 6, gg = ln >> 4, n = ln & 15, l = n >> 1;
        const int w = ks < 4 * NBW ? 16 * (ks >> 2) + 4 * gg + (ks & 3) : 16 * NBW + gg;
        float v = 0.f;
        if (l < m3 && w < D3) {
            const float2 t = p.tw3[(unsigned)(l * w) % (unsigned)D3];
            v = ((n & 1) ? -t.y : t.x) * p.scale * (p.herm ? herm_weight(l, D3) : 1.0f);
        }
        sTwA[e] = v;
    }
    for (int e = tid; e < g.U * 4 * MT2 * 64; e += nthreads) {
        const int ln = e & 63, mt = (e >> 6) % MT2, ur = (e >> 6) / MT2;
        const int mk = 16 * mt + 4 * (ln & 3) + ((ln & 15) >> 2);         // kappa index of A-operand row ln & 15: accumulator row 4 g + r <-> 4 r + g
        const int i = 16 * (ur >> 2) + 4 * (ln >> 4) + (ur & 3);            // slot of k-index g in k-step (u, r)
        float2 v = make_float2(0.f, 0.f);
        if (mk < m2 && i < g.nslot2) {
            v = p.tw2[(unsigned)((2 * mk + 1) * i) % (unsigned)(2 * D2)];
            if (i == 0) v.y = (D2 & 1) ? 0.f : ((mk & 1) ? -1.f : 1.f);
        }
        sTwB[e] = v;
    }
    for (int e = tid; e < g.U * 16; e += nthreads) sTwist2[e] = p.tw2[min(e, 2 * D2 - 1)];
    for (int e = tid; e < g.nks1 * g.MT1 * 64; e += nthreads) {
        const int ln = e & 63, mt = (e >> 6) % g.MT1, ks = (e >> 6) / g.MT1;
        const int mk = 16 * mt + (ln & 15), i = 4 * ks + (ln >> 4);
        float2 v = make_float2(0.f, 0.f);
        if (mk < m1 && i < g.nslot1) {
            v = p.tw1[(unsigned)((2 * mk + 1) * i) % (unsigned)(2 * D1)];
            if (i == 0) v.y = (D1 & 1) ? 0.f : ((mk & 1) ? -1.f : 1.f);
        }
        sTw1[e] = v;
    }

    __syncthreads();

    f32x4 Xa_p[MT2], Xa_m[MT2];                                               // spectrum of the slot's first plane
    f32x4 C[MT2], S[MT2];
    for (int q = 0; q < n_units; ++q) {
        const int u = q % U, second = (q / U) & 1, slot = wave + VOL_WAVES * (q / (2 * U));
        cP = nP; cQ = nQ;
        if (q + 1 < n_units) issue(unit_plane(q + 1), (q + 1) % U);
        if (u == 0) {
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) { C[mt] = f32x4{0, 0, 0, 0}; S[mt] = f32x4{0, 0, 0, 0}; }
        }
        // T axis: rows of P and Q
        f32x4 TP = f32x4{0, 0, 0, 0}, TQ = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NBW; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tw = sTwA[(4 * c + e) * 64 + lane];
                TP = mfma16(__uint_as_float(cP.w[c][e]), tw, TP);
                TQ = mfma16(__uint_as_float(cQ.w[c][e]), tw, TQ);
            }
        if (NARROW) {
            const float tw = sTwA[(4 * NBW) * 64 + lane];
            TP = mfma16(__uint_as_float(cP.nrw), tw, TP);
            TQ = mfma16(__uint_as_float(cQ.nrw), tw, TQ);
        }
        // twist + pair, then the dim2 stage
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 t = sTwist2[16 * u + 4 * gq + r];
            const float sum = TP[r] + TQ[r], dif = TP[r] - TQ[r];
            float Dp = t.x * sum + t.y * (sg * vol_xor1(dif));
            float Ep = t.x * dif + t.y * (sg * vol_xor1(sum));
            if (r == 0) {
                const float jq = sg * vol_xor1(TQ[0]);
                if (u == 0 && gq == 0) { Dp = TP[0]; Ep = even2 ? jq : 0.f; }
            }
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const float2 tb = sTwB[((u * 4 + r) * MT2 + mt) * 64 + lane];
                C[mt] = mfma16(tb.x, Dp, C[mt]);
                S[mt] = mfma16(tb.y, Ep, S[mt]);
            }
        }
        if (u != U - 1) continue;
        // plane spectrum: +kappa = C - i S, -kappa = C + i S
        if (!second) {
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float js = sg * vol_xor1(S[mt][r]);
                    Xa_p[mt][r] = C[mt][r] - js;
                    Xa_m[mt][r] = C[mt][r] + js;
                }
            continue;
        }
        // second plane of the slot: twist + pair along dim1, to LDS
        const float2 t1 = p.tw1[slot];
        const bool n_ok = n16 < 2 * m3;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float js = sg * vol_xor1(S[mt][r]);
                const float bp = C[mt][r] - js, bm = C[mt][r] + js;
                const int mk = 16 * mt + 4 * r + gq;
#pragma unroll
                for (int sgn = 0; sgn < 2; ++sgn) {
                    const float a = sgn ? Xa_m[mt][r] : Xa_p[mt][r], b = sgn ? bm : bp;
                    const float sum = a + b, dif = a - b;
                    float Dp = t1.x * sum + t1.y * (sg * vol_xor1(dif));
                    float Ep = t1.x * dif + t1.y * (sg * vol_xor1(sum));
                    const float jb = sg * vol_xor1(b);
                    if (slot == 0) { Dp = a; Ep = even1 ? jb : 0.f; }
                    const int j2 = sgn ? 2 * m2 - 1 - mk : mk;
                    if (mk < m2 && n_ok) {
                        const int at = slot * g.RP + j2 * 2 * m3 + n16;
                        sD[at] = Dp;
                        sE[at] = Ep;
                    }
                }
            }
    }
    __syncthreads();

    // ---- phase 2: leading axis out of LDS, blocks of 16 complex columns dealt to the waves.  A lane owns one complex column
    // (re and im are two MFMA column tiles), so i S needs no lane exchange and a row of the result leaves as 128 contiguous bytes.
    float2* out = reinterpret_cast<float2*>(p.out) + (size_t)vol * (size_t)(4 * m1 * m2 * m3);            // 4 corners x m1 m2 m3 complex
    for (int blk = wave; blk < g.NT; blk += VOL_WAVES) {
        const int c = 16 * blk + n16;                                          // complex column j2 * m3 + l
        const bool cvalid = 2 * c < g.C2;
        const int j2 = c / m3, l = c - j2 * m3;
        const int cc = j2 >= m2, jj2 = j2 - cc * m2;
        for (int mt = 0; mt < g.MT1; ++mt) {
            f32x4 Cr = f32x4{0, 0, 0, 0}, Ci = f32x4{0, 0, 0, 0}, Sr = f32x4{0, 0, 0, 0}, Si = f32x4{0, 0, 0, 0};
            for (int ks = 0; ks < g.nks1; ++ks) {
                const int row = min(4 * ks + gq, g.nslot1 - 1);
                const float2 bD = *reinterpret_cast<const float2*>(sD + row * g.RP + 32 * blk + 2 * n16);
                const float2 bE = *reinterpret_cast<const float2*>(sE + row * g.RP + 32 * blk + 2 * n16);
                const float2 tw = sTw1[(ks * g.MT1 + mt) * 64 + lane];
                Cr = mfma16(tw.x, bD.x, Cr);
                Ci = mfma16(tw.x, bD.y, Ci);
                Sr = mfma16(tw.y, bE.x, Sr);
                Si = mfma16(tw.y, bE.y, Si);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mk = 16 * mt + 4 * gq + r;
                if (cvalid && mk < m1) {
                    // corner-major (4, m1, m2, m3): corner = (j1 >= m1) + 2 (j2 >= m2);  +kappa = C - i S,  -kappa = C + i S
                    const size_t lo = (((size_t)(0 + 2 * cc) * m1 + mk) * m2 + jj2) * m3 + l;
                    const size_t hi = (((size_t)(1 + 2 * cc) * m1 + (m1 - 1 - mk)) * m2 + jj2) * m3 + l;
                    out[lo] = make_float2(Cr[r] + Si[r], Ci[r] - Sr[r]);
                    out[hi] = make_float2(Cr[r] - Si[r], Ci[r] + Sr[r]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- K3v
// The identities of the header backwards.  Leading axis (phase 1, 16-column tiles of the volume's truncated spectrum dealt to the
// waves):  Ek = O[+kappa] + O[-kappa], Dk = O[+kappa] - O[-kappa],  Pc[i] = sum cos(theta_i) Ek,  Qs[i] = sum sin(theta_i) i Dk,
// y[i] = Pc + Qs, y[N - i] = Qs - Pc (slot 0: y[0] = Pc, y[N/2] = Qs), plane = conj(t) y  ->  per-plane spectra in LDS.
// Planes (phase 2, dealt to the waves): the same along dim2 with the plane spectrum as the A operand (M = T-mode column), so that
// the result is directly the B operand of the T-axis stage and a lane ends up with four consecutive output columns of a row.
struct VolInvShape {
    int nslot1, nslot2, MTS, U, NWT, nk1, nk2, C2, NT, RP;
    size_t lds;
};

static VolInvShape vol_inv_shape(int D1, int D2, int D3, int m1, int m2, int m3) {
    VolInvShape g;
    g.nslot1 = (D1 + 1) / 2;
    g.nslot2 = (D2 + 1) / 2;
    g.MTS = (g.nslot1 + 15) / 16;
    g.U = (g.nslot2 + 15) / 16;
    g.NWT = (D3 + 15) / 16;
    g.nk1 = (m1 + 3) / 4;
    g.nk2 = (m2 + 3) / 4;
    g.C2 = 2 * m2 * 2 * m3;
    g.NT = (g.C2 + 31) / 32;               // blocks of 16 complex columns
    g.RP = 32 * g.NT + 8;                   // = 8 mod 16: the two 4-row groups of a half-wave's 8-byte phase-1 writes fall on distinct banks
    g.lds = (size_t)D1 * g.RP * 4 + (size_t)g.nk1 * g.MTS * 64 * 8 + (size_t)g.MTS * 16 * 8 + (size_t)g.nk2 * g.U * 64 * 8;
    return g;
}

bool vol3d_inv_applies(int n_vol, int D1, int D2, int D3, int m1, int m2, int m3) {
    if (n_vol < VOL_MIN_VOLUMES) return false;
    if (D1 < 4 || D2 < 4 || D3 < 2 || D1 > 64 || D2 > 64 || D3 > 32) return false;
    if (2 * m1 > D1 || 2 * m2 > D2 || m1 > 32 || m2 > 32 || 2 * m3 > 16 || m3 > D3 / 2 + 1) return false;
    if ((long long)D1 * D2 * D3 * 4 >= (1LL << 31)) return false;
    return vol_inv_shape(D1, D2, D3, m1, m2, m3).lds <= VOL_LDS_LIMIT;
}

template <int MTS, int U, int NWT>
__global__ __launch_bounds__(64 * VOL_WAVES) void dft3d_inv_volume_kernel(Vol3dParams p, VolInvShape g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D1 = p.D1, D2 = p.D2, D3 = p.D3, m1 = p.m1, m2 = p.m2, m3 = p.m3;
    float* sZ = reinterpret_cast<float*>(smem);                                  // [D1][RP]: per-plane truncated spectra
    float2* sTw1 = reinterpret_cast<float2*>(sZ + (size_t)D1 * g.RP);            // [nk1][MTS][64]  A operand, phase 1
    float2* sTwist1 = sTw1 + g.nk1 * MTS * 64;                                   // [16 MTS]
    float2* sTw2 = sTwist1 + 16 * MTS;                                           // [nk2][U][64]    B operand, phase 2
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = 64 * VOL_WAVES;
    const float sg = (lane & 1) ? 1.f : -1.f;
    const bool even1 = !(D1 & 1), even2 = !(D2 & 1);

    for (int e = tid; e < g.nk1 * MTS * 64; e += nthreads) {
        const int ln = e & 63, mt = (e >> 6) % MTS, ks = (e >> 6) / MTS;
        const int i = 16 * mt + (ln & 15), mk = 4 * ks + (ln >> 4);
        float2 v = make_float2(0.f, 0.f);
        if (mk < m1 && i < g.nslot1) {
            v = p.tw1[(unsigned)((2 * mk + 1) * i) % (unsigned)(2 * D1)];
            if (i == 0) v.y = even1 ? ((mk & 1) ? -1.f : 1.f) : 0.f;
        }
        sTw1[e] = v;
    }
    for (int e = tid; e < 16 * MTS; e += nthreads) sTwist1[e] = p.tw1[min(e, 2 * D1 - 1)];
    for (int e = tid; e < g.nk2 * U * 64; e += nthreads) {
        const int ln = e & 63, u = (e >> 6) % U, ks = (e >> 6) / U;
        const int i = 16 * u + (ln & 15), mk = 4 * ks + (ln >> 4);
        float2 v = make_float2(0.f, 0.f);
        if (mk < m2 && i < g.nslot2) {
            v = p.tw2[(unsigned)((2 * mk + 1) * i) % (unsigned)(2 * D2)];
            if (i == 0) v.y = even2 ? ((mk & 1) ? -1.f : 1.f) : 0.f;
        }
        sTw2[e] = v;
    }
    // T-axis weights of this lane (A operand of the last stage): G[w = 16 wt + n16][n = 4 g + r], scale and Hermitian weight folded in
    float G[NWT][4];
#pragma unroll
    for (int wt = 0; wt < NWT; ++wt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int w = 16 * wt + n16, n = 4 * gq + r, l = n >> 1;
            float v = 0.f;
            if (l < m3 && w < D3) {
                const float2 t = p.tw3[(unsigned)(l * w) % (unsigned)D3];
                v = ((n & 1) ? -t.y : t.x) * p.scale * (p.herm ? herm_weight(l, D3) : 1.0f);
            }
            G[wt][r] = v;
        }
    // dim2 twist of this lane's slot (phase 2: slot 16 u + n16)
    float c2[U], s2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float2 t = p.tw2[min(16 * u + n16, 2 * D2 - 1)];
        c2[u] = t.x; s2[u] = t.y;
    }
    __syncthreads();

    // ---- phase 1: leading axis.  A lane owns one complex column of the volume's spectrum (16 of them per block): 8-byte loads,
    // i Dk and the untwist without lane exchange, 8-byte LDS writes.
    const int vol = blockIdx.x;
    const float2* O = reinterpret_cast<const float2*>(p.in) + (size_t)vol * (size_t)(4 * m1 * m2 * m3);
    const size_t cstride = (size_t)m1 * m2 * m3;                                 // complex elements per corner
    for (int blk = wave; blk < g.NT; blk += VOL_WAVES) {
        const int c = min(16 * blk + n16, g.C2 / 2 - 1);
        const int j2 = c / m3, l = c - j2 * m3;
        const int cc = j2 >= m2, jj2 = j2 - cc * m2;
        const float2* Olo = O + (size_t)(2 * cc) * cstride + (size_t)jj2 * m3 + l;          // + row * m2 * m3
        const float2* Ohi = Olo + cstride;
        const int rstride = m2 * m3;
        f32x4 Pr[MTS], Pi[MTS], Qr[MTS], Qi[MTS];
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) { Pr[mt] = f32x4{0, 0, 0, 0}; Pi[mt] = f32x4{0, 0, 0, 0}; Qr[mt] = f32x4{0, 0, 0, 0}; Qi[mt] = f32x4{0, 0, 0, 0}; }
        auto fetch = [&](int ks, float2& lo, float2& hi) {
            const int mk = min(4 * ks + gq, m1 - 1);
            lo = Olo[(size_t)mk * rstride];
            hi = Ohi[(size_t)(m1 - 1 - mk) * rstride];
        };
        float2 lo, hi, nlo, nhi;
        fetch(0, nlo, nhi);
        for (int ks = 0; ks < g.nk1; ++ks) {
            lo = nlo; hi = nhi;
            if (ks + 1 < g.nk1) fetch(ks + 1, nlo, nhi);
            const float er = lo.x + hi.x, ei = lo.y + hi.y;                      // Ek
            const float jr = -(lo.y - hi.y), ji = lo.x - hi.x;                   // i Dk
#pragma unroll
            for (int mt = 0; mt < MTS; ++mt) {
                const float2 tw = sTw1[(ks * MTS + mt) * 64 + lane];
                Pr[mt] = mfma16(tw.x, er, Pr[mt]);
                Pi[mt] = mfma16(tw.x, ei, Pi[mt]);
                Qr[mt] = mfma16(tw.y, jr, Qr[mt]);
                Qi[mt] = mfma16(tw.y, ji, Qi[mt]);
            }
        }
        const int colf = 32 * blk + 2 * n16;
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * mt + 4 * gq + r;
                const float2 t = sTwist1[i];
                const float ar = Pr[mt][r] + Qr[mt][r], ai = Pi[mt][r] + Qi[mt][r];
                const float br = Pr[mt][r] - Qr[mt][r], bi = Pi[mt][r] - Qi[mt][r];
                // plane i = conj(t) (Pc + Qs),  plane N - i = -t (Qs - Pc) = t (Pc - Qs)
                float2 zi = make_float2(t.x * ar + t.y * ai, t.x * ai - t.y * ar);
                float2 zn = make_float2(t.x * br - t.y * bi, t.x * bi + t.y * br);
                if (i == 0) { zi = make_float2(Pr[mt][r], Pi[mt][r]); zn = make_float2(Qi[mt][r], -Qr[mt][r]); }      // planes 0 and N/2: Pc, -i Qs
                if (i < g.nslot1) {
                    *reinterpret_cast<float2*>(sZ + i * g.RP + colf) = zi;
                    if (i > 0) *reinterpret_cast<float2*>(sZ + (D1 - i) * g.RP + colf) = zn;
                    else if (even1) *reinterpret_cast<float2*>(sZ + (D1 / 2) * g.RP + colf) = zn;
                }
            }
    }
    __syncthreads();

    // ---- phase 2: planes
    float* ybase = p.out + (size_t)vol * D1 * D2 * D3;
    for (int d1 = wave; d1 < D1; d1 += VOL_WAVES) {
        const float* Z = sZ + d1 * g.RP;
        f32x4 Pc2[U], Qs2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { Pc2[u] = f32x4{0, 0, 0, 0}; Qs2[u] = f32x4{0, 0, 0, 0}; }
        for (int ks = 0; ks < g.nk2; ++ks) {
            const int mk = min(4 * ks + gq, m2 - 1);
            const float lo = Z[mk * 2 * m3 + n16], hi = Z[(2 * m2 - 1 - mk) * 2 * m3 + n16];
            const float ek = lo + hi, jd = sg * vol_xor1(lo - hi);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float2 tw = sTw2[(ks * U + u) * 64 + lane];
                Pc2[u] = mfma16(ek, tw.x, Pc2[u]);
                Qs2[u] = mfma16(jd, tw.y, Qs2[u]);
            }
        }
        float* yplane = ybase + (size_t)d1 * D2 * D3;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = 16 * u + n16;
            const f32x4 a = Pc2[u] + Qs2[u], b = Pc2[u] - Qs2[u];
            // registers (0, 1) and (2, 3) are (re, im) of T-modes 2 g and 2 g + 1:  i v = (-im, re)
            f32x4 UP, UQ;
            UP[0] = c2[u] * a[0] + s2[u] * a[1];  UP[1] = c2[u] * a[1] - s2[u] * a[0];
            UP[2] = c2[u] * a[2] + s2[u] * a[3];  UP[3] = c2[u] * a[3] - s2[u] * a[2];
            UQ[0] = c2[u] * b[0] - s2[u] * b[1];  UQ[1] = c2[u] * b[1] + s2[u] * b[0];
            UQ[2] = c2[u] * b[2] - s2[u] * b[3];  UQ[3] = c2[u] * b[3] + s2[u] * b[2];
            if (i == 0) {
                UP = Pc2[u];
                UQ[0] = Qs2[u][1]; UQ[1] = -Qs2[u][0]; UQ[2] = Qs2[u][3]; UQ[3] = -Qs2[u][2];
            }
            const bool p_ok = i < g.nslot2, q_ok = p_ok && (i > 0 || even2);
            const int hP = i, hQ = i > 0 ? D2 - i : D2 / 2;
#pragma unroll
            for (int wt = 0; wt < NWT; ++wt) {
                f32x4 YP = f32x4{0, 0, 0, 0}, YQ = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    YP = mfma16(G[wt][r], UP[r], YP);
                    YQ = mfma16(G[wt][r], UQ[r], YQ);
                }
                const int w0 = 16 * wt + 4 * gq;
                auto put = [&](const f32x4& Y, int h) {
                    float* row = yplane + (size_t)h * D3;
                    if (w0 + 3 < D3) {
                        *reinterpret_cast<f4u*>(row + w0) = f4u{{Y[0], Y[1], Y[2], Y[3]}};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (w0 + e < D3) row[w0 + e] = Y[e];
                    }
                };
                if (p_ok) put(YP, hP);
                if (q_ok) put(YQ, hQ);
            }
        }
    }
}

template <int MT2, int NBW, bool NARROW>
static int launch_fwd_volume_t(const Vol3dParams& p, const VolShape& g, hipStream_t s) {
    static int lds_slot[64];
    const void* k = reinterpret_cast<const void*>(dft3d_fwd_volume_kernel<MT2, NBW, NARROW>);
    if (!ensure_dynamic_lds(k, g.lds, lds_slot)) { set_error("dft3d_fwd_volume: cannot raise the dynamic LDS limit to %zu", g.lds); return -5; }
    {
        char name[64];
        snprintf(name, sizeof(name), "uno::dft3d_fwd_volume_kernel<%d, %d, %s>", MT2, NBW, NARROW ? "true" : "false");
        ProfScope prof(name, (double)p.n_vol * ((double)p.D1 * p.D2 * p.D3 * 4.0 + 8.0 * p.m1 * p.m2 * p.m3 * 4.0), s);
        hipLaunchKernelGGL((dft3d_fwd_volume_kernel<MT2, NBW, NARROW>), dim3(p.n_vol), dim3(64 * VOL_WAVES), g.lds, s, p, g);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft3d_fwd_volume launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_dft3d_fwd_volume(const Vol3dParams& p, hipStream_t s) {
    const VolShape g = vol_shape(p.D1, p.D2, p.D3, p.m1, p.m2, p.m3);
#define UNO_CASE(a, b, c) if (g.MT2 == a && g.NBW == b && g.NARROW == c) return launch_fwd_volume_t<a, b, (c != 0)>(p, g, s);
    UNO_CASE(1, 1, 0) UNO_CASE(1, 1, 1) UNO_CASE(1, 2, 0) UNO_CASE(1, 2, 1)
    UNO_CASE(2, 1, 0) UNO_CASE(2, 1, 1) UNO_CASE(2, 2, 0) UNO_CASE(2, 2, 1)
#undef UNO_CASE
    set_error("dft3d_fwd_volume: unsupported tile configuration");
    return -2;
}

template <int MTS, int U, int NWT>
static int launch_inv_volume_t(const Vol3dParams& p, const VolInvShape& g, hipStream_t s) {
    static int lds_slot[64];
    const void* k = reinterpret_cast<const void*>(dft3d_inv_volume_kernel<MTS, U, NWT>);
    if (!ensure_dynamic_lds(k, g.lds, lds_slot)) { set_error("dft3d_inv_volume: cannot raise the dynamic LDS limit to %zu", g.lds); return -5; }
    {
        char name[64];
        snprintf(name, sizeof(name), "uno::dft3d_inv_volume_kernel<%d, %d, %d>", MTS, U, NWT);
        ProfScope prof(name, (double)p.n_vol * ((double)p.D1 * p.D2 * p.D3 * 4.0 + 8.0 * p.m1 * p.m2 * p.m3 * 4.0), s);
        hipLaunchKernelGGL((dft3d_inv_volume_kernel<MTS, U, NWT>), dim3(p.n_vol), dim3(64 * VOL_WAVES), g.lds, s, p, g);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft3d_inv_volume launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_dft3d_inv_volume(const Vol3dParams& p, hipStream_t s) {
    const VolInvShape g = vol_inv_shape(p.D1, p.D2, p.D3, p.m1, p.m2, p.m3);
#define UNO_CASE(a, b, c) if (g.MTS == a && g.U == b && g.NWT == c) return launch_inv_volume_t<a, b, c>(p, g, s);
    UNO_CASE(1, 1, 1) UNO_CASE(1, 1, 2) UNO_CASE(1, 2, 1) UNO_CASE(1, 2, 2)
    UNO_CASE(2, 1, 1) UNO_CASE(2, 1, 2) UNO_CASE(2, 2, 1) UNO_CASE(2, 2, 2)
#undef UNO_CASE
    set_error("dft3d_inv_volume: unsupported tile configuration");
    return -2;
}

}  // namespace uno
