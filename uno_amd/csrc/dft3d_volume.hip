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
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

namespace uno {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr size_t VOL_LDS_LIMIT = 160 * 1024;
constexpr int VOL_MIN_VOLUMES = 48;     // one workgroup per volume; measured (tools/dev/vol3dtime.py, widths 8 and 16): still ahead of the plane path at 64 volumes

#ifdef UNO_VOL_DEV
// development build (tools/dev/mkvariant.py voldev dft3d_volume.hip -DUNO_VOL_DEV): cycle stamps per wave at the phase boundaries,
// buffer address from UNO_VOL_STAMPS; knock-outs by UNO_VOL_EXP.  Nothing of this is in the product build.
__device__ unsigned long long* g_vol_stamps = nullptr;
__device__ int g_vol_exp = 0;
#define VOL_STAMP(i_) do { if (g_vol_stamps && lane == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_vol_stamps[((size_t)blockIdx.x * 16 + wave) * 8 + (i_)] = __builtin_readcyclecounter(); } } while (0)
#define VOL_STAMP_NOWAIT(i_) do { if (g_vol_stamps && lane == 0) { g_vol_stamps[((size_t)blockIdx.x * 16 + wave) * 8 + (i_)] = __builtin_readcyclecounter(); } } while (0)
static void vol_dev_setup(const char* which) {
    const char* e = getenv("UNO_VOL_STAMPS");
    const char* w = getenv("UNO_VOL_WHICH");
    if (w && strcmp(w, which) != 0) e = nullptr;
    unsigned long long* ptr = e ? reinterpret_cast<unsigned long long*>((uintptr_t)strtoull(e, nullptr, 0)) : nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_vol_stamps), &ptr, sizeof(ptr));
    const char* x = getenv("UNO_VOL_EXP");
    int xv = x ? atoi(x) : 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_vol_exp), &xv, sizeof(xv));
}
#else
#define VOL_STAMP(i_) do {} while (0)
#define VOL_STAMP_NOWAIT(i_) do {} while (0)
#endif

__device__ __forceinline__ float vol_xor1(float v) {       // the value held by lane ^ 1 (DPP quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// ---- host-built operand tables (one per device and shape; built on first use with hipMalloc + a synchronous copy, like the
// twiddle tables: a shape must have been seen once before a stream capture records it - GraphedStep's warm-up steps do that)
struct VolTabKey {
    int dev, kind, D1, D2, D3, m1, m2, m3, herm;
    unsigned scale_bits;
    bool operator<(const VolTabKey& o) const {
        return std::tie(dev, kind, D1, D2, D3, m1, m2, m3, herm, scale_bits) < std::tie(o.dev, o.kind, o.D1, o.D2, o.D3, o.m1, o.m2, o.m3, o.herm, o.scale_bits);
    }
};
static const float* vol_table(const VolTabKey& key_in, const std::function<void(std::vector<float>&)>& build) {
    static std::mutex mu;
    static std::map<VolTabKey, float*> cache;
    VolTabKey key = key_in;
    if (hipGetDevice(&key.dev) != hipSuccess) { set_error("hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    std::vector<float> host;
    build(host);
    float* d = static_cast<float*>(upload_table(host.data(), host.size() * sizeof(float)));
    if (!d) {
        set_error("dft3d volume operand table: allocation of %zu bytes failed: %s", host.size() * sizeof(float), hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    cache[key] = d;
    return d;
}
static inline float host_herm_weight(int l, int N) { return (l == 0 || 2 * l == N) ? 1.0f : 2.0f; }

constexpr int VF_WAVES = 8;              // K1v: 2 waves per SIMD, 256 registers each (constants + a whole slot of rows in flight)

struct VolShape {
    int nslot1, nslot2;      // pair slots along dim1 / dim2: slot 0 = (0, N/2), slot i = (i, N - i)
    int U;                   // 16-slot tile pairs of a plane
    int MT2, MT1;            // 16-row tiles of kappa along dim2 / dim1
    int NBW, NARROW;         // dim3 in 16-column blocks: NBW with a 16-byte piece per lane (4 k-steps), + one 4-byte block (1 k-step)
    int KA;                  // k-steps of the T-axis stage
    int C2;                  // floats of a per-plane truncated spectrum: 2 m2 rows x m3 complex
    int RP;                  // its row pitch in LDS (floats), = 32 mod 64 (the two k-slots of a half-wave's 8-byte B-operand reads fall on
                             // distinct banks) and > C2: the first pad float of a row takes the stores of lanes outside the spectrum
    int NT;                  // 32-column (16 complex) blocks of C2
    int nks1;                // k-steps of the leading-axis stage
    int NREG;                // per-lane register constants (floats), padded to a multiple of 4
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
    while ((g.RP & 63) != 32 || g.RP <= g.C2) g.RP += 32;
    g.nks1 = (g.nslot1 + 3) / 4;
    g.NREG = (g.KA + 8 * g.U * g.MT2 + 8 * g.U + 3) & ~3;
    g.lds = (size_t)2 * g.nslot1 * g.RP * 4                      // D' / E' rows, interleaved per slot
            + (size_t)g.nks1 * g.MT1 * 64 * 8;                   // dim1 (cos, sin) (A operand of the leading-axis stage)
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

// Operand tables of K1v.  Register constants of lane ln, element k at [(k / 4) * 64 + ln] * 4 + k % 4 (a lane reads its constants as
// 16-byte pieces, a wave 1 KB per piece), k counting  twA[KA] | twB[U][4][MT2] (cos, sin) | twist[U][4] (cos, sin * sg);  then the
// LDS image of the leading-axis A operand [nks1][MT1][64] (cos, sin).
static const float* vol_fwd_table(const Vol3dParams& p, const VolShape& g) {
    VolTabKey key{0, 0, p.D1, p.D2, p.D3, p.m1, p.m2, p.m3, p.herm, 0u};
    memcpy(&key.scale_bits, &p.scale, 4);
    return vol_table(key, [&](std::vector<float>& t) {
        const int D1 = p.D1, D2 = p.D2, D3 = p.D3, m1 = p.m1, m2 = p.m2, m3 = p.m3, U = g.U, MT2 = g.MT2, NBW = g.NBW;
        t.assign((size_t)g.NREG * 64 + (size_t)g.nks1 * g.MT1 * 64 * 2, 0.f);
        auto reg = [&](int k, int ln) -> float& { return t[((size_t)(k >> 2) * 64 + ln) * 4 + (k & 3)]; };
        for (int ln = 0; ln < 64; ++ln) {
            const int gg = ln >> 4, n = ln & 15, l = n >> 1;
            int k = 0;
            for (int ks = 0; ks < g.KA; ++ks, ++k) {
                const int w = ks < 4 * NBW ? 16 * (ks >> 2) + 4 * gg + (ks & 3) : 16 * NBW + gg;
                float v = 0.f;
                if (l < m3 && w < D3) {
                    const float2 tw = twiddle_value((long long)l * w % D3, D3);
                    v = ((n & 1) ? -tw.y : tw.x) * p.scale * (p.herm ? host_herm_weight(l, D3) : 1.0f);
                }
                reg(k, ln) = v;
            }
            for (int u = 0; u < U; ++u)
                for (int r = 0; r < 4; ++r)
                    for (int mt = 0; mt < MT2; ++mt, k += 2) {
                        const int mk = 16 * mt + 4 * (ln & 3) + ((ln & 15) >> 2);     // kappa index of A-operand row ln & 15: accumulator row 4 g + r <-> 4 r + g
                        const int i = 16 * u + 4 * (ln >> 4) + r;                    // slot of k-index g in k-step (u, r)
                        float2 v = make_float2(0.f, 0.f);
                        if (mk < m2 && i < g.nslot2) {
                            v = twiddle_value((long long)(2 * mk + 1) * i % (2 * D2), 2 * D2);
                            if (i == 0) v.y = (D2 & 1) ? 0.f : ((mk & 1) ? -1.f : 1.f);
                        }
                        reg(k, ln) = v.x; reg(k + 1, ln) = v.y;
                    }
            const float sg = (ln & 1) ? 1.f : -1.f;
            for (int u = 0; u < U; ++u)
                for (int r = 0; r < 4; ++r, k += 2) {
                    const float2 v = twiddle_value(std::min(16 * u + 4 * gg + r, 2 * D2 - 1), 2 * D2);
                    reg(k, ln) = v.x; reg(k + 1, ln) = v.y * sg;
                }
        }
        float* t1 = t.data() + (size_t)g.NREG * 64;
        for (int e = 0; e < g.nks1 * g.MT1 * 64; ++e) {
            const int ln = e & 63, mt = (e >> 6) % g.MT1, ks = (e >> 6) / g.MT1;
            const int mk = 16 * mt + (ln & 15), i = 4 * ks + (ln >> 4);
            float2 v = make_float2(0.f, 0.f);
            if (mk < m1 && i < g.nslot1) {
                v = twiddle_value((long long)(2 * mk + 1) * i % (2 * D1), 2 * D1);
                if (i == 0) v.y = (D1 & 1) ? 0.f : ((mk & 1) ? -1.f : 1.f);
            }
            t1[2 * e] = v.x; t1[2 * e + 1] = v.y;
        }
    });
}

// ---------------------------------------------------------------------------------------------------------------- K1v
// What the counters and a co-issue probe (tools/probes/coissue_probe.hip) said about the first form of this kernel: on a gfx950 SIMD
// a wave's VALU instructions do NOT overlap ANOTHER wave's MFMAs (the older wave's queued MFMA holds the issue stage: time = sum
// of the two), so with 4 waves per SIMD the plane phase took the SUM of its MFMA cycles (144 x 32 per wave) and its ~1 200 VALU /
// 770 SALU instructions per wave - not their maximum - and the waves of a SIMD finished 22 / 28 / 34 / 40 k cycles after the
// barrier (oldest first).  So this form removes instructions instead of adding waves: every twiddle operand of the T-axis and dim2
// stages is a per-lane REGISTER constant read once from a host-built table (no LDS reads or modulo arithmetic in the loop), the
// units of a slot (2 planes x U tile pairs) are unrolled by position (no copies, no run-time stage selection), a position's rows for
// the NEXT slot are requested as soon as its MFMAs have consumed the current ones (a whole slot = 10 KB per wave in flight), and the
// pair / twist results go to LDS through precomputed offsets with no lane masks.
//
// One plane of the volume -> its truncated spectrum in registers: rows +kappa / -kappa with kappa index 16 mt + 4 r + g
// (g = lane >> 4), column n = lane & 15 (re / im of T-mode n >> 1 interleaved).
template <int MT2, int NBW, bool NARROW, int U>
__global__ __launch_bounds__(64 * VF_WAVES) void dft3d_fwd_volume_kernel(Vol3dParams p, VolShape g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D1 = p.D1, D2 = p.D2, D3 = p.D3, m1 = p.m1, m2 = p.m2, m3 = p.m3;
    float* sDE = reinterpret_cast<float*>(smem);                             // [nslot1][D' | E'][RP]
    float2* sTw1 = reinterpret_cast<float2*>(sDE + (size_t)2 * g.nslot1 * g.RP);      // [nks1][MT1][64]
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int KA = 4 * NBW + (NARROW ? 1 : 0);
    constexpr int NREG = (KA + 8 * U * MT2 + 8 * U + 3) & ~3;
    VOL_STAMP_NOWAIT(0);

    const int vol = blockIdx.x;
    const size_t vol_elems = (size_t)D1 * D2 * D3;
    const float* vbase = p.in + (size_t)vol * vol_elems;
    const size_t after = ((size_t)(p.n_vol - vol)) * vol_elems * 4;          // bytes from this volume to the end of the tensor
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)std::min<size_t>(after, 0x7fffffffu), 0x00020000);
    const float sg = (lane & 1) ? 1.f : -1.f;
    const bool even2 = !(D2 & 1), even1 = !(D1 & 1);

    // rows of tile pair u: P = slot 16 u + n16, Q = its partner D2 - slot (slot 0: D2 / 2); rows past the last slot: clamped, their
    // twiddles are zero.  Byte offsets of this lane's pieces inside a plane:
    auto row_off = [&](int u, bool q) -> unsigned {
        int i = min(16 * u + n16, g.nslot2 - 1);
        int h = q ? (i == 0 ? D2 / 2 : D2 - i) : i;
        return (unsigned)(h * D3 + 4 * gq) * 4u;
    };
    unsigned offP[U], offQ[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { offP[u] = row_off(u, false); offQ[u] = row_off(u, true); }

    struct Piece { u32x4v w[NBW]; unsigned nrw; };
    Piece LP[2 * U], LQ[2 * U];                                               // one register set per unit position of a slot
    const unsigned plane_bytes = (unsigned)(D2 * D3) * 4u;
    auto plane_of = [&](int slot, int second) -> int {
        if (!second) return slot;
        return slot == 0 ? D1 / 2 : D1 - slot;                              // (odd D1, slot 0: plane D1 / 2 is loaded and ignored)
    };
    auto issue = [&](Piece& P_, Piece& Q_, int plane, int u) {
        const unsigned sbase = (unsigned)plane * plane_bytes;
#pragma unroll
        for (int c = 0; c < NBW; ++c) {
            P_.w[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, offP[u] + 64u * c, sbase, 0);
            Q_.w[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, offQ[u] + 64u * c, sbase, 0);
        }
        if (NARROW) {
            // column 16 NBW + g: this lane's piece starts at column 4 g -> + (16 NBW - 3 g) columns
            P_.nrw = __builtin_amdgcn_raw_buffer_load_b32(rsrc, offP[u] + 4u * (16 * NBW - 3 * gq), sbase, 0);
            Q_.nrw = __builtin_amdgcn_raw_buffer_load_b32(rsrc, offQ[u] + 4u * (16 * NBW - 3 * gq), sbase, 0);
        }
    };

    // this wave's slots: wave, wave + VF_WAVES, ...; the rows of the first one are on their way while the constants arrive
    const int my_slots = max((g.nslot1 - wave + VF_WAVES - 1) / VF_WAVES, 0);
    if (my_slots > 0) {
#pragma unroll
        for (int pos = 0; pos < 2 * U; ++pos) issue(LP[pos], LQ[pos], plane_of(wave, pos / U), pos % U);
    }

    // ---- constants: registers from the host-built table, the leading-axis operand into LDS
    float creg[NREG];
    {
        const f32x4* ct = reinterpret_cast<const f32x4*>(p.ctab) + lane;
#pragma unroll
        for (int k4 = 0; k4 < NREG / 4; ++k4) {
            const f32x4 v = ct[k4 * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) creg[4 * k4 + e] = v[e];
        }
        const f32x4* t1 = reinterpret_cast<const f32x4*>(p.ctab + (size_t)NREG * 64);
        f32x4* s1 = reinterpret_cast<f32x4*>(sTw1);
        for (int e = tid; e < g.nks1 * g.MT1 * 32; e += 64 * VF_WAVES) s1[e] = t1[e];
    }
    auto twA = [&](int ks) -> float { return creg[ks]; };
    auto twB = [&](int u, int r, int mt) -> float2 { const int k = KA + 2 * ((u * 4 + r) * MT2 + mt); return make_float2(creg[k], creg[k + 1]); };
    auto tws = [&](int u, int r) -> float2 { const int k = KA + 8 * U * MT2 + 2 * (u * 4 + r); return make_float2(creg[k], creg[k + 1]); };

    // LDS offsets (floats, inside a slot's D' row; E' = + RP) of the +kappa / -kappa rows this lane holds; lanes outside the spectrum
    // (kappa >= m2, column >= 2 m3) write the row's first pad float
    int offPl[MT2][4], offMi[MT2][4];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mk = 16 * mt + 4 * r + gq;
            const bool ok = mk < m2 && n16 < 2 * m3;
            offPl[mt][r] = ok ? mk * 2 * m3 + n16 : g.C2;
            offMi[mt][r] = ok ? (2 * m2 - 1 - mk) * 2 * m3 + n16 : g.C2;
        }
    VOL_STAMP_NOWAIT(1);
    VOL_STAMP_NOWAIT(2);

    // ---- phase 1: planes -> paired, twisted per-plane spectra in LDS.
    // Program order inside a wave is what lets VALU work run under the MFMAs (see above), so the stages of consecutive unit positions
    // are interleaved in the source: while position pos is twisted and goes through its dim2 stage, the T-axis MFMAs of position
    // pos + 1 (at the last position: of the next slot's first) are already issued, and the plane epilogue of position pos - 1 (VALU
    // + LDS stores) sits between them.  REM = slots this wave still has, this one included (1, 2, or 3 = more).
    auto t_axis = [&](const Piece& P_, const Piece& Q_, f32x4& TP_, f32x4& TQ_) {
        TP_ = f32x4{0, 0, 0, 0}; TQ_ = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NBW; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tw = twA(4 * c + e);
                TP_ = mfma16(__uint_as_float(P_.w[c][e]), tw, TP_);
                TQ_ = mfma16(__uint_as_float(Q_.w[c][e]), tw, TQ_);
            }
        if (NARROW) {
            const float tw = twA(4 * NBW);
            TP_ = mfma16(__uint_as_float(P_.nrw), tw, TP_);
            TQ_ = mfma16(__uint_as_float(Q_.nrw), tw, TQ_);
        }
    };
    // (the variants with the most constants and rows in flight - two tile pairs with two column blocks or two kappa tiles - have no
    // registers left for the look-ahead: they run the stages of a position in sequence)
    constexpr bool AHEAD = !(U == 2 && NBW + MT2 > 2);
    f32x4 TP, TQ;                                                             // T-axis results of the position about to be processed
    auto slot_body = [&](int slot, auto rem, auto is_slot0) {
        constexpr int REM = decltype(rem)::value;
        constexpr bool SLOT0 = decltype(is_slot0)::value;                    // planes 0 and D1 / 2
        const float2 t1 = p.tw1[slot];
        const float t1ys = t1.y * sg;
        float* rowD = sDE + (size_t)slot * 2 * g.RP;
        f32x4 C[MT2], S[MT2], Xa_p[MT2], Xa_m[MT2];
        // plane epilogues (called one position late)
        auto first_plane = [&]() {
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float js = sg * vol_xor1(S[mt][r]);
                    Xa_p[mt][r] = C[mt][r] - js;
                    Xa_m[mt][r] = C[mt][r] + js;
                }
        };
        auto second_plane = [&]() {                                          // twist + pair along dim1, to LDS
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float js = sg * vol_xor1(S[mt][r]);
                    const float bp = C[mt][r] - js, bm = C[mt][r] + js;
#pragma unroll
                    for (int sgn = 0; sgn < 2; ++sgn) {
                        const float a = sgn ? Xa_m[mt][r] : Xa_p[mt][r], b = sgn ? bm : bp;
                        float Dp, Ep;
                        if (SLOT0) {
                            Dp = a; Ep = even1 ? sg * vol_xor1(b) : 0.f;
                        } else {
                            const float sum = a + b, dif = a - b;
                            Dp = t1.x * sum + t1ys * vol_xor1(dif);
                            Ep = t1.x * dif + t1ys * vol_xor1(sum);
                        }
                        const int at = sgn ? offMi[mt][r] : offPl[mt][r];
                        rowD[at] = Dp;
                        rowD[at + g.RP] = Ep;
                    }
                }
        };
#pragma unroll
        for (int pos = 0; pos < 2 * U; ++pos) {
            const int u = pos % U;
            // the T-axis stage one position ahead, and the rows that position needs one slot later
            f32x4 TPn = f32x4{0, 0, 0, 0}, TQn = f32x4{0, 0, 0, 0};
            if (!AHEAD) {
                t_axis(LP[pos], LQ[pos], TP, TQ);
                if (REM >= 2) issue(LP[pos], LQ[pos], plane_of(slot + VF_WAVES, pos / U), u);
            } else if (pos + 1 < 2 * U) {
                t_axis(LP[pos + 1], LQ[pos + 1], TPn, TQn);
                if (REM >= 2) issue(LP[pos + 1], LQ[pos + 1], plane_of(slot + VF_WAVES, (pos + 1) / U), (pos + 1) % U);
            } else if (REM >= 2) {
                t_axis(LP[0], LQ[0], TPn, TQn);
                if (REM >= 3) issue(LP[0], LQ[0], slot + 2 * VF_WAVES, 0);
            }
            // the epilogue of the plane the previous position completed
            if (pos > 0 && (pos - 1) % U == U - 1) first_plane();
            if (u == 0) {
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt) { C[mt] = f32x4{0, 0, 0, 0}; S[mt] = f32x4{0, 0, 0, 0}; }
            }
            // twist + pair, then the dim2 stage
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 t = tws(u, r);                                   // (cos, sin * sg)
                const float sum = TP[r] + TQ[r], dif = TP[r] - TQ[r];
                float Dp = t.x * sum + t.y * vol_xor1(dif);
                float Ep = t.x * dif + t.y * vol_xor1(sum);
                if (r == 0 && u == 0) {
                    const float jq = sg * vol_xor1(TQ[0]);
                    if (gq == 0) { Dp = TP[0]; Ep = even2 ? jq : 0.f; }
                }
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt) {
                    const float2 tb = twB(u, r, mt);
                    C[mt] = mfma16(tb.x, Dp, C[mt]);
                    S[mt] = mfma16(tb.y, Ep, S[mt]);
                }
            }
            if (AHEAD) { TP = TPn; TQ = TQn; }
        }
        second_plane();
    };
    // this wave's first slot: its first position's T-axis stage, and that position's rows for the second slot
    if (AHEAD && my_slots > 0) {
        t_axis(LP[0], LQ[0], TP, TQ);
        if (my_slots > 1) issue(LP[0], LQ[0], wave + VF_WAVES, 0);
    }
    for (int q = 0; q < my_slots; ++q) {
        const int slot = wave + VF_WAVES * q, remaining = my_slots - q;
        if (slot == 0) {
            if (remaining >= 3) slot_body(slot, std::integral_constant<int, 3>{}, std::true_type{});
            else if (remaining == 2) slot_body(slot, std::integral_constant<int, 2>{}, std::true_type{});
            else slot_body(slot, std::integral_constant<int, 1>{}, std::true_type{});
        } else if (remaining >= 3) slot_body(slot, std::integral_constant<int, 3>{}, std::false_type{});
        else if (remaining == 2) slot_body(slot, std::integral_constant<int, 2>{}, std::false_type{});
        else slot_body(slot, std::integral_constant<int, 1>{}, std::false_type{});
    }
    VOL_STAMP(3);
    __syncthreads();
    VOL_STAMP_NOWAIT(4);

    // ---- phase 2: leading axis out of LDS, blocks of 16 complex columns dealt to the waves.  A lane owns one complex column
    // (re and im are two MFMA column tiles), so i S needs no lane exchange and a row of the result leaves as 128 contiguous bytes.
    // (results leave through raw buffer stores: columns / rows outside the spectrum carry an out-of-range offset and are dropped)
    float2* out = reinterpret_cast<float2*>(p.out) + (size_t)vol * (size_t)(4 * m1 * m2 * m3);            // 4 corners x m1 m2 m3 complex
    constexpr unsigned OOB = 0xC0000000u;
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const unsigned rowbytes = (unsigned)(m2 * m3) * 8u, cbytes = (unsigned)m1 * rowbytes;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)(4u * cbytes), 0x00020000);
    for (int blk = wave; blk < g.NT; blk += VF_WAVES) {
        const int c = 16 * blk + n16;                                          // complex column j2 * m3 + l
        const bool cvalid = 2 * c < g.C2;
        const int j2 = c / m3, l = c - j2 * m3;
        const int cc = j2 >= m2, jj2 = j2 - cc * m2;
        // corner-major (4, m1, m2, m3): corner = (j1 >= m1) + 2 (j2 >= m2)
        const unsigned colb = cvalid ? (unsigned)(2 * cc) * cbytes + (unsigned)(jj2 * m3 + l) * 8u : OOB;
        const float* pD = sDE + 32 * blk + 2 * n16;
        for (int mt = 0; mt < g.MT1; ++mt) {
            f32x4 Cr = f32x4{0, 0, 0, 0}, Ci = f32x4{0, 0, 0, 0}, Sr = f32x4{0, 0, 0, 0}, Si = f32x4{0, 0, 0, 0};
            auto fetch = [&](int ks, float2& bD, float2& bE, float2& tw) {
                const int row = min(4 * ks + gq, g.nslot1 - 1);
                bD = *reinterpret_cast<const float2*>(pD + (size_t)row * 2 * g.RP);
                bE = *reinterpret_cast<const float2*>(pD + (size_t)row * 2 * g.RP + g.RP);
                tw = sTw1[(ks * g.MT1 + mt) * 64 + lane];
            };
            float2 bD, bE, tw, nD, nE, ntw;
            fetch(0, nD, nE, ntw);
#pragma unroll 2
            for (int ks = 0; ks < g.nks1; ++ks) {
                bD = nD; bE = nE; tw = ntw;
                fetch(min(ks + 1, g.nks1 - 1), nD, nE, ntw);                  // the next step's operands are on their way during the MFMAs
                Cr = mfma16(tw.x, bD.x, Cr);
                Ci = mfma16(tw.x, bD.y, Ci);
                Sr = mfma16(tw.y, bE.x, Sr);
                Si = mfma16(tw.y, bE.y, Si);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mk = 16 * mt + 4 * gq + r;
                // +kappa = C - i S,  -kappa = C + i S
                const unsigned lo = mk < m1 ? colb + (unsigned)mk * rowbytes : OOB;
                const unsigned hi = mk < m1 ? colb + cbytes + (unsigned)(m1 - 1 - mk) * rowbytes : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2v{__float_as_uint(Cr[r] + Si[r]), __float_as_uint(Ci[r] - Sr[r])}, ors, lo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2v{__float_as_uint(Cr[r] - Si[r]), __float_as_uint(Ci[r] + Sr[r])}, ors, hi, 0, 0);
            }
        }
    }
    VOL_STAMP_NOWAIT(5);
    VOL_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------- K3v
// The identities of the header backwards.  Leading axis (phase 1, 16-column tiles of the volume's truncated spectrum dealt to the
// waves):  Ek = O[+kappa] + O[-kappa], Dk = O[+kappa] - O[-kappa],  Pc[i] = sum cos(theta_i) Ek,  Qs[i] = sum sin(theta_i) i Dk,
// y[i] = Pc + Qs, y[N - i] = Qs - Pc (slot 0: y[0] = Pc, y[N/2] = Qs), plane = conj(t) y  ->  per-plane spectra in LDS.
// Planes (phase 2, dealt to the waves): the same along dim2 with the plane spectrum as the A operand (M = T-mode column), so that
// the result is directly the B operand of the T-axis stage and a lane ends up with four consecutive output columns of a row.
constexpr int VI_WAVES = 8;

struct VolInvShape {
    int nslot1, nslot2, MTS, U, NWT, nk1, nk2, NK2, C2, NT, RP, NREG;
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
    g.NK2 = (g.nk2 + 1) & ~1;               // compiled k-step counts of the dim2 stage: 2, 4, 6, 8 (the padded step multiplies zeros)
    g.C2 = 2 * m2 * 2 * m3;
    g.NT = (g.C2 + 31) / 32;               // blocks of 16 complex columns
    g.RP = 32 * g.NT + 8;                   // = 8 mod 16: the two 4-row groups of a half-wave's 8-byte phase-1 writes fall on distinct banks
    g.NREG = (2 * g.NK2 * g.U + 4 * g.NWT + 2 * g.U + 3) & ~3;
    g.lds = (size_t)D1 * g.RP * 4 + (size_t)g.nk1 * g.MTS * 64 * 8 + (size_t)g.MTS * 16 * 8;
    return g;
}

bool vol3d_inv_applies(int n_vol, int D1, int D2, int D3, int m1, int m2, int m3) {
    if (n_vol < VOL_MIN_VOLUMES) return false;
    if (D1 < 4 || D2 < 4 || D3 < 2 || D1 > 64 || D2 > 64 || D3 > 32) return false;
    if (2 * m1 > D1 || 2 * m2 > D2 || m1 > 32 || m2 > 32 || 2 * m3 > 16 || m3 > D3 / 2 + 1) return false;
    if ((long long)D1 * D2 * D3 * 4 >= (1LL << 31)) return false;
    return vol_inv_shape(D1, D2, D3, m1, m2, m3).lds <= VOL_LDS_LIMIT;
}

// Operand tables of K3v: register constants (layout as K1v's)  tw2[NK2][U] (cos, sin) | G[NWT][4] | twist2[U] (cos, sin);  then the LDS
// image  sTw1[nk1][MTS][64] (cos, sin) | sTwist1[16 MTS] (cos, sin).
static const float* vol_inv_table(const Vol3dParams& p, const VolInvShape& g) {
    VolTabKey key{0, 1, p.D1, p.D2, p.D3, p.m1, p.m2, p.m3, p.herm, 0u};
    memcpy(&key.scale_bits, &p.scale, 4);
    return vol_table(key, [&](std::vector<float>& t) {
        const int D1 = p.D1, D2 = p.D2, D3 = p.D3, m1 = p.m1, m2 = p.m2, m3 = p.m3, U = g.U, MTS = g.MTS;
        const bool even1 = !(D1 & 1), even2 = !(D2 & 1);
        t.assign((size_t)g.NREG * 64 + (size_t)g.nk1 * MTS * 64 * 2 + (size_t)MTS * 16 * 2, 0.f);
        auto reg = [&](int k, int ln) -> float& { return t[((size_t)(k >> 2) * 64 + ln) * 4 + (k & 3)]; };
        for (int ln = 0; ln < 64; ++ln) {
            const int n16 = ln & 15, gq = ln >> 4;
            int k = 0;
            for (int ks = 0; ks < g.NK2; ++ks)
                for (int u = 0; u < U; ++u, k += 2) {
                    const int i = 16 * u + n16, mk = 4 * ks + gq;
                    float2 v = make_float2(0.f, 0.f);
                    if (mk < m2 && i < g.nslot2) {
                        v = twiddle_value((long long)(2 * mk + 1) * i % (2 * D2), 2 * D2);
                        if (i == 0) v.y = even2 ? ((mk & 1) ? -1.f : 1.f) : 0.f;
                    }
                    reg(k, ln) = v.x; reg(k + 1, ln) = v.y;
                }
            // T-axis weights (A operand of the last stage): G[w = 16 wt + n16][n = 4 g + r], scale and Hermitian weight folded in
            for (int wt = 0; wt < g.NWT; ++wt)
                for (int r = 0; r < 4; ++r, ++k) {
                    const int w = 16 * wt + n16, n = 4 * gq + r, l = n >> 1;
                    float v = 0.f;
                    if (l < m3 && w < D3) {
                        const float2 tw = twiddle_value((long long)l * w % D3, D3);
                        v = ((n & 1) ? -tw.y : tw.x) * p.scale * (p.herm ? host_herm_weight(l, D3) : 1.0f);
                    }
                    reg(k, ln) = v;
                }
            for (int u = 0; u < U; ++u, k += 2) {
                const float2 v = twiddle_value(std::min(16 * u + n16, 2 * D2 - 1), 2 * D2);
                reg(k, ln) = v.x; reg(k + 1, ln) = v.y;
            }
        }
        float* t1 = t.data() + (size_t)g.NREG * 64;
        for (int e = 0; e < g.nk1 * MTS * 64; ++e) {
            const int ln = e & 63, mt = (e >> 6) % MTS, ks = (e >> 6) / MTS;
            const int i = 16 * mt + (ln & 15), mk = 4 * ks + (ln >> 4);
            float2 v = make_float2(0.f, 0.f);
            if (mk < m1 && i < g.nslot1) {
                v = twiddle_value((long long)(2 * mk + 1) * i % (2 * D1), 2 * D1);
                if (i == 0) v.y = even1 ? ((mk & 1) ? -1.f : 1.f) : 0.f;
            }
            t1[2 * e] = v.x; t1[2 * e + 1] = v.y;
        }
        float* tt = t1 + (size_t)g.nk1 * MTS * 64 * 2;
        for (int e = 0; e < 16 * MTS; ++e) {
            const float2 v = twiddle_value(std::min(e, 2 * D1 - 1), 2 * D1);
            tt[2 * e] = v.x; tt[2 * e + 1] = v.y;
        }
    });
}

// 8 waves (two per SIMD): see K1v - the plane phase is bound by the SUM of a SIMD's MFMA and VALU cycles, so the dim2 / T-axis
// operands are per-lane register constants from a host-built table and the k loop of the dim2 stage is compiled (NK2 steps).
// Output rows leave through raw buffer stores whose per-lane offsets carry the validity (a lane with nothing to store points past the
// end of the volume: the hardware drops the access), so the plane loop is ONE basic block - with the guards as branches every store
// sat between two exec-mask branches and each MFMA chain's latency was exposed four times per plane.  A4: D3 % 4 == 0 (a lane's
// 4-column piece is whole or absent; otherwise the last column tile stores element by element).
template <int MTS, int U, int NWT, int NK2, bool A4>
__global__ __launch_bounds__(64 * VI_WAVES) void dft3d_inv_volume_kernel(Vol3dParams p, VolInvShape g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D1 = p.D1, D2 = p.D2, D3 = p.D3, m1 = p.m1, m2 = p.m2, m3 = p.m3;
    float* sZ = reinterpret_cast<float*>(smem);                                  // [D1][RP]: per-plane truncated spectra
    float2* sTw1 = reinterpret_cast<float2*>(sZ + (size_t)D1 * g.RP);            // [nk1][MTS][64]  A operand, phase 1
    float2* sTwist1 = sTw1 + g.nk1 * MTS * 64;                                   // [16 MTS]
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = 64 * VI_WAVES;
    const float sg = (lane & 1) ? 1.f : -1.f;
    const bool even1 = !(D1 & 1), even2 = !(D2 & 1);
    constexpr int NREG = (2 * NK2 * U + 4 * NWT + 2 * U + 3) & ~3;
    VOL_STAMP_NOWAIT(0);

    // phase-1 tables -> LDS (16-byte copies of the host-built image; sTw1 and sTwist1 are contiguous there and here)
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.ctab + (size_t)NREG * 64);
        f32x4* dst = reinterpret_cast<f32x4*>(sTw1);
        const int n4 = (g.nk1 * MTS * 64 + 16 * MTS) / 2;
        for (int e = tid; e < n4; e += nthreads) dst[e] = src[e];
    }
    float creg[NREG];
    {
        const f32x4* ct = reinterpret_cast<const f32x4*>(p.ctab) + lane;
#pragma unroll
        for (int k4 = 0; k4 < NREG / 4; ++k4) {
            const f32x4 v = ct[k4 * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) creg[4 * k4 + e] = v[e];
        }
    }
    auto tw2 = [&](int ks, int u) -> float2 { const int k = 2 * (ks * U + u); return make_float2(creg[k], creg[k + 1]); };
    auto G = [&](int wt, int r) -> float { return creg[2 * NK2 * U + 4 * wt + r]; };
    auto twist2 = [&](int u) -> float2 { const int k = 2 * NK2 * U + 4 * NWT + 2 * u; return make_float2(creg[k], creg[k + 1]); };
    VOL_STAMP_NOWAIT(1);
    __syncthreads();
    VOL_STAMP_NOWAIT(2);

    // ---- phase 1: leading axis.  A lane owns one complex column of the volume's spectrum (16 of them per block): 8-byte loads,
    // i Dk and the untwist without lane exchange, 8-byte LDS writes.
    const int vol = blockIdx.x;
    const float2* O = reinterpret_cast<const float2*>(p.in) + (size_t)vol * (size_t)(4 * m1 * m2 * m3);
    const size_t cstride = (size_t)m1 * m2 * m3;                                 // complex elements per corner
    int rowI[MTS][4], rowN[MTS][4];                                              // LDS rows (float offsets) of slot i and its partner; < 0: none
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * mt + 4 * gq + r;
            rowI[mt][r] = i < g.nslot1 ? i * g.RP : -1;
            rowN[mt][r] = i >= g.nslot1 ? -1 : i > 0 ? (D1 - i) * g.RP : even1 ? (D1 / 2) * g.RP : -1;
        }
    for (int blk = wave; blk < g.NT; blk += VI_WAVES) {
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
                if (mt == 0 && r == 0 && gq == 0) { zi = make_float2(Pr[mt][r], Pi[mt][r]); zn = make_float2(Qi[mt][r], -Qr[mt][r]); }      // planes 0 and N/2: Pc, -i Qs
                // rows past the last slot (and the partner of slot 0 on an odd axis) go to the pad floats of row 0
                *reinterpret_cast<float2*>(sZ + (rowI[mt][r] >= 0 ? rowI[mt][r] + colf : 32 * g.NT)) = zi;
                *reinterpret_cast<float2*>(sZ + (rowN[mt][r] >= 0 ? rowN[mt][r] + colf : 32 * g.NT + 2)) = zn;
            }
    }
    VOL_STAMP(3);
    __syncthreads();
    VOL_STAMP_NOWAIT(4);

    // ---- phase 2: planes.  LDS offsets of this lane's +kappa / -kappa values per k-step (rows past m2: clamped, their twiddles are zero)
    int zlo[NK2], zhi[NK2];
#pragma unroll
    for (int ks = 0; ks < NK2; ++ks) {
        const int mk = min(4 * ks + gq, m2 - 1);
        zlo[ks] = mk * 2 * m3 + n16;
        zhi[ks] = (2 * m2 - 1 - mk) * 2 * m3 + n16;
    }
    // store plan: byte offsets (inside a plane) of this lane's 4-column pieces of rows P / Q of tile pair u; absent rows / columns -> OOB
    constexpr unsigned OOB = 0xC0000000u;
    const size_t vol_bytes = (size_t)D1 * D2 * D3 * 4;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)vol * D1 * D2 * D3), 0, (int)vol_bytes, 0x00020000);
    unsigned oP[U], oQ[U], oPl[U], oQl[U];
    const int wl = 16 * (NWT - 1) + 4 * gq;                                      // first column of the piece in the last tile
    const int nlast = min(max(D3 - wl, 0), 4);                                   // its valid columns
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = 16 * u + n16;
        const bool p_ok = i < g.nslot2, q_ok = p_ok && (i > 0 || even2);
        const int hP = i, hQ = i > 0 ? D2 - i : D2 / 2;
        oP[u] = p_ok ? (unsigned)(hP * D3 + 4 * gq) * 4u : OOB;
        oQ[u] = q_ok ? (unsigned)(hQ * D3 + 4 * gq) * 4u : OOB;
        oPl[u] = (p_ok && nlast > 0) ? oP[u] + 64u * (NWT - 1) : OOB;
        oQl[u] = (q_ok && nlast > 0) ? oQ[u] + 64u * (NWT - 1) : OOB;
    }
    const unsigned plane_bytes = (unsigned)(D2 * D3) * 4u;
    auto put = [&](const f32x4& Y, unsigned off, unsigned offl, int wt, unsigned sbase) {
        u32x4v d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __float_as_uint(Y[e]);
        if (wt < NWT - 1) {
            __builtin_amdgcn_raw_buffer_store_b128(d, yrs, off + 64u * wt, sbase, 0);
        } else if (A4) {
            __builtin_amdgcn_raw_buffer_store_b128(d, yrs, offl, sbase, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b32(d[e], yrs, e < nlast ? offl + 4u * e : OOB, sbase, 0);
        }
    };
    // The three stages of a plane (dim2 GEMM -> untwist -> T-axis GEMM + stores) depend on each other, and a wave's VALU work only runs
    // under MFMAs that precede it in ITS OWN program order (see K1v): so the loop is software-pipelined across planes - while plane j goes
    // through its T-axis stage, the LDS reads, the dim2 stage and the untwist of plane j + 1 are interleaved with it in the source.
    struct Plane { f32x4 UP[U], UQ[U]; };
    auto read_rows = [&](int d1, float (&lo)[NK2], float (&hi)[NK2]) {
        const float* Z = sZ + d1 * g.RP;
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks) { lo[ks] = Z[zlo[ks]]; hi[ks] = Z[zhi[ks]]; }
    };
    auto dim2_stage = [&](const float (&lo)[NK2], const float (&hi)[NK2], f32x4 (&Pc2)[U], f32x4 (&Qs2)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) { Pc2[u] = f32x4{0, 0, 0, 0}; Qs2[u] = f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks) {
            const float ek = lo[ks] + hi[ks], jd = sg * vol_xor1(lo[ks] - hi[ks]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float2 tw = tw2(ks, u);
                Pc2[u] = mfma16(ek, tw.x, Pc2[u]);
                Qs2[u] = mfma16(jd, tw.y, Qs2[u]);
            }
        }
    };
    auto untwist = [&](const f32x4 (&Pc2)[U], const f32x4 (&Qs2)[U], Plane& pl) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float2 t2 = twist2(u);
            const float c2 = t2.x, s2 = t2.y;
            const f32x4 a = Pc2[u] + Qs2[u], b = Pc2[u] - Qs2[u];
            // registers (0, 1) and (2, 3) are (re, im) of T-modes 2 g and 2 g + 1:  i v = (-im, re)
            f32x4 UP, UQ;
            UP[0] = c2 * a[0] + s2 * a[1];  UP[1] = c2 * a[1] - s2 * a[0];
            UP[2] = c2 * a[2] + s2 * a[3];  UP[3] = c2 * a[3] - s2 * a[2];
            UQ[0] = c2 * b[0] - s2 * b[1];  UQ[1] = c2 * b[1] + s2 * b[0];
            UQ[2] = c2 * b[2] - s2 * b[3];  UQ[3] = c2 * b[3] + s2 * b[2];
            if (u == 0 && n16 == 0) {
                UP = Pc2[u];
                UQ[0] = Qs2[u][1]; UQ[1] = -Qs2[u][0]; UQ[2] = Qs2[u][3]; UQ[3] = -Qs2[u][2];
            }
            pl.UP[u] = UP; pl.UQ[u] = UQ;
        }
    };
    auto t_stage = [&](const Plane& pl, int u, unsigned sbase) {
#pragma unroll
        for (int wt = 0; wt < NWT; ++wt) {
            f32x4 YP = f32x4{0, 0, 0, 0}, YQ = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                YP = mfma16(G(wt, r), pl.UP[u][r], YP);
                YQ = mfma16(G(wt, r), pl.UQ[u][r], YQ);
            }
            put(YP, oP[u], oPl[u], wt, sbase);
            put(YQ, oQ[u], oQl[u], wt, sbase);
        }
    };
    const int my_planes = max((D1 - wave + VI_WAVES - 1) / VI_WAVES, 0);
    Plane cur;
    if (my_planes > 0) {
        float lo[NK2], hi[NK2];
        f32x4 Pc2[U], Qs2[U];
        read_rows(wave, lo, hi);
        dim2_stage(lo, hi, Pc2, Qs2);
        untwist(Pc2, Qs2, cur);
    }
    for (int j = 0; j + 1 < my_planes; ++j) {
        const int d1 = wave + VI_WAVES * j;
        const unsigned sbase = (unsigned)d1 * plane_bytes;
        float lo[NK2], hi[NK2];
        f32x4 Pc2[U], Qs2[U];
        Plane nxt;
        read_rows(d1 + VI_WAVES, lo, hi);
        t_stage(cur, 0, sbase);
        dim2_stage(lo, hi, Pc2, Qs2);
        if (U > 1) t_stage(cur, U - 1, sbase);
        untwist(Pc2, Qs2, nxt);
        cur = nxt;
        // the interleave, spelled out for the scheduler (it clusters the MFMAs and leaves the untwist behind them otherwise):
        // row reads | T-axis MFMAs of tile pair 0 with the sums / differences of the next plane | its dim2 MFMAs | T-axis MFMAs of tile
        // pair 1 with its untwist
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NK2, 0);
#pragma unroll
        for (int i = 0; i < 8 * NWT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 2 * NK2 * U; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (U > 1) {
#pragma unroll
            for (int i = 0; i < 8 * NWT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
    }
    if (my_planes > 0) {
        const unsigned sbase = (unsigned)(wave + VI_WAVES * (my_planes - 1)) * plane_bytes;
#pragma unroll
        for (int u = 0; u < U; ++u) t_stage(cur, u, sbase);
    }
    VOL_STAMP_NOWAIT(5);
    VOL_STAMP(6);
}

template <int MT2, int NBW, bool NARROW, int U>
static int launch_fwd_volume_t(const Vol3dParams& p, const VolShape& g, hipStream_t s) {
    static int lds_slot[64];
    const void* k = reinterpret_cast<const void*>(dft3d_fwd_volume_kernel<MT2, NBW, NARROW, U>);
    if (!ensure_dynamic_lds(k, g.lds, lds_slot)) { set_error("dft3d_fwd_volume: cannot raise the dynamic LDS limit to %zu", g.lds); return -5; }
    {
        char name[64];
        snprintf(name, sizeof(name), "uno::dft3d_fwd_volume_kernel<%d, %d, %s, %d>", MT2, NBW, NARROW ? "true" : "false", U);
        ProfScope prof(name, (double)p.n_vol * ((double)p.D1 * p.D2 * p.D3 * 4.0 + 8.0 * p.m1 * p.m2 * p.m3 * 4.0), s);
        hipLaunchKernelGGL((dft3d_fwd_volume_kernel<MT2, NBW, NARROW, U>), dim3(p.n_vol), dim3(64 * VF_WAVES), g.lds, s, p, g);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft3d_fwd_volume launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_dft3d_fwd_volume(const Vol3dParams& p_in, hipStream_t s) {
    const VolShape g = vol_shape(p_in.D1, p_in.D2, p_in.D3, p_in.m1, p_in.m2, p_in.m3);
    Vol3dParams p = p_in;
    p.ctab = vol_fwd_table(p, g);
    if (!p.ctab) return -6;
#ifdef UNO_VOL_DEV
    vol_dev_setup("fwd");
#endif
#define UNO_CASE(a, b, c, d) if (g.MT2 == a && g.NBW == b && g.NARROW == c && g.U == d) return launch_fwd_volume_t<a, b, (c != 0), d>(p, g, s);
    UNO_CASE(1, 1, 0, 1) UNO_CASE(1, 1, 1, 1) UNO_CASE(1, 2, 0, 1) UNO_CASE(1, 2, 1, 1)
    UNO_CASE(2, 1, 0, 1) UNO_CASE(2, 1, 1, 1) UNO_CASE(2, 2, 0, 1) UNO_CASE(2, 2, 1, 1)
    UNO_CASE(1, 1, 0, 2) UNO_CASE(1, 1, 1, 2) UNO_CASE(1, 2, 0, 2) UNO_CASE(1, 2, 1, 2)
    UNO_CASE(2, 1, 0, 2) UNO_CASE(2, 1, 1, 2) UNO_CASE(2, 2, 0, 2) UNO_CASE(2, 2, 1, 2)
#undef UNO_CASE
    set_error("dft3d_fwd_volume: unsupported tile configuration");
    return -2;
}

template <int MTS, int U, int NWT, int NK2, bool A4>
static int launch_inv_volume_t(const Vol3dParams& p, const VolInvShape& g, hipStream_t s) {
    static int lds_slot[64];
    const void* k = reinterpret_cast<const void*>(dft3d_inv_volume_kernel<MTS, U, NWT, NK2, A4>);
    if (!ensure_dynamic_lds(k, g.lds, lds_slot)) { set_error("dft3d_inv_volume: cannot raise the dynamic LDS limit to %zu", g.lds); return -5; }
    {
        char name[64];
        snprintf(name, sizeof(name), "uno::dft3d_inv_volume_kernel<%d, %d, %d, %d, %s>", MTS, U, NWT, NK2, A4 ? "true" : "false");
        ProfScope prof(name, (double)p.n_vol * ((double)p.D1 * p.D2 * p.D3 * 4.0 + 8.0 * p.m1 * p.m2 * p.m3 * 4.0), s);
        hipLaunchKernelGGL((dft3d_inv_volume_kernel<MTS, U, NWT, NK2, A4>), dim3(p.n_vol), dim3(64 * VI_WAVES), g.lds, s, p, g);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft3d_inv_volume launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_dft3d_inv_volume(const Vol3dParams& p_in, hipStream_t s) {
    const VolInvShape g = vol_inv_shape(p_in.D1, p_in.D2, p_in.D3, p_in.m1, p_in.m2, p_in.m3);
    Vol3dParams p = p_in;
    p.ctab = vol_inv_table(p, g);
    if (!p.ctab) return -6;
#ifdef UNO_VOL_DEV
    vol_dev_setup("inv");
#endif
    const bool a4 = (p.D3 & 3) == 0;
#define UNO_CASE2(a, b, c, d) if (g.NK2 == d) return a4 ? launch_inv_volume_t<a, b, c, d, true>(p, g, s) : launch_inv_volume_t<a, b, c, d, false>(p, g, s);
#define UNO_CASE(a, b, c) if (g.MTS == a && g.U == b && g.NWT == c) { UNO_CASE2(a, b, c, 2) UNO_CASE2(a, b, c, 4) UNO_CASE2(a, b, c, 6) UNO_CASE2(a, b, c, 8) }
    UNO_CASE(1, 1, 1) UNO_CASE(1, 1, 2) UNO_CASE(1, 2, 1) UNO_CASE(1, 2, 2)
    UNO_CASE(2, 1, 1) UNO_CASE(2, 1, 2) UNO_CASE(2, 2, 1) UNO_CASE(2, 2, 2)
#undef UNO_CASE
#undef UNO_CASE2
    set_error("dft3d_inv_volume: unsupported tile configuration");
    return -2;
}

}  // namespace uno
