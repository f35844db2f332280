// K1-B / K3-B - the pruned 2-D transforms for bfloat16 IMAGES with the row stage (the transform along W: ~85 % of the flops) on
// v_mfma_f32_16x16x32_bf16 (BASELINE.json configs[4]: bf16 activations; reference integral_operators.py:187, 206 run rfft2 / irfft2).
//
// Why: the bf16-image forms of K1 / K3 (dft2d_fwd_kernel.h, dft2d_inv_kernel.h) widen every element to f32 and run the f32 MFMA
// row stage of the f32 kernels - 16x fewer flops per cycle than the bf16 pipe, and at 32 modes those kernels are bound by MFMA
// cycles (C5 block round 3: 0.23 / 0.155 of the HBM roofline; the mixed backward SLOWER than f32).  With bf16 images the image
// operand of the row-stage GEMM is EXACT in bf16, so only the constant twiddle operand needs care: it is split into hi + lo bf16
// (16 significant bits, relative error 2^-17 - two orders below the bf16 rounding of the result) and the stage costs two bf16
// MFMAs per (16 x 32) x (32 x 16) block instead of eight f32 ones.  Accumulation stays f32; the column stage (along H, ~1/8 of the
// flops, K = 16 per tile) stays on the f32 MFMA and is the code of K1-HT / K3.
//
// Forward (K1-B): T[h][n] = sum_w x[h][w] B[w][n], n = (cos | sin) x modes - NOT the symmetric (x[w] +- x[W-w]) form of the f32
//   kernels: the sums / differences of two bf16 values are not bf16 values, and the bf16 pipe has the cycles to spare.
//   * A operand = the image itself: lane (row i = lane & 15, k-group g = lane >> 4) loads the 16 bytes x[i][32 s + 8 g .. + 7]
//     straight from global memory into the operand registers (no LDS, no VALU); a wave owns 32-row tiles (two operand tiles per
//     k-step share every twiddle read) and keeps eight k-steps = 16 KB in flight;
//   * B operand = twiddles in operand layout in LDS, tabulated for ONE chunk of KC = 256 columns only (a 1024-column table with
//     hi + lo for 64 output columns would be 256 KB).  Chunks are walked from the LAST to the first and the accumulator is rotated
//     between chunks by rho[l] = exp(-2 pi i l KC / W) (Horner: T = S_0 + rho (S_1 + rho (S_2 + ...)), 8 f32 FMAs per accumulator
//     register pair and chunk) - no second accumulator set.
// Inverse (K3-B): y[h][w] = sum_l Ur[h][l] cos(2 pi l w / W) - Ui[h][l] sin(2 pi l w / W) with U from the column stage (f32, K3's
//   +-k paired form, accumulators = operand registers of the row stage: k-slot (g, j) <-> mode 4 j' + g exactly as in K3):
//   * B operand = U rotated to the chunk (U rho'_c, f32 VALU), split hi + lo; A operand = the chunk-0 twiddles hi + lo in LDS;
//     three products (hi hi, lo hi, hi lo): relative error ~2^-16; a wave owns 32-row tiles: every twiddle read feeds two tiles;
//   * a lane ends with four consecutive columns of one row; 32 rows x NC columns are staged in LDS and leave as whole row
//     segments, 16 bytes per lane.
// What the first version measured (C5 size, 256 images of 1024^2, knock-outs in one gpurun call): with 16-row tiles the kernels took
// 148 / 190 us; without any MFMA AND without HBM traffic still 93 / 99 us - the twiddle-operand reads (8 KB of LDS per KB of image),
// their waits and the per-k-step bookkeeping were the pole, not the matrix pipe (45 % busy) and not the memory system; 14 / 12 us
// were table construction per workgroup (now host-built tables, copied); non-temporal loads re-fetched the half lines of the
// operand-layout access pattern (446^2: 134 -> 105 us with the default policy).
#include "uno_common.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace uno {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_b16(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two f32 -> (hi, lo) bf16 pairs: v = hi + lo + O(2^-17 v)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = bf16_pack2(a, b);
    const float ra = a - __uint_as_float(hi << 16);
    const float rb = b - __uint_as_float(hi & 0xffff0000u);
    lo = bf16_pack2(ra, rb);
}

// development: per-phase cycle stamps of one wave per workgroup (Dft2dParams.exp & 16; the counters go to the buffer p.rowfreq points at)
__device__ __forceinline__ unsigned long long b16_clock() { return __builtin_readcyclecounter(); }
#define B16_STAMP(acc) do { if constexpr ((xp & 16) != 0) { const unsigned long long t_ = b16_clock(); (acc) += t_ - t_prev; t_prev = t_; } } while (0)

constexpr int B16_WAVES = 8;            // waves per workgroup (two per SIMD)
constexpr int B16_KSC = 8;              // forward: k-steps (of 32 columns) per chunk = per table

static int b16_cu_count() {
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return usable_cus(cus[dev]);
}

static int b16_exp(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// ---------------------------------------------------------------------------------------------------------------- host tables
// The twiddle operands in MFMA operand layout, hi / lo bf16, built once per (device, direction, W, modes, chunk) in double
// precision and kept on the device: a workgroup's prologue is a 16-byte-per-thread copy instead of 16 k modulo + gather + split
// sequences (measured: 12-14 us per launch at the C5 size, one workgroup per CU - nothing to hide it behind).
static unsigned short host_bf16(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float host_widen(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void host_cossin(long long idx, int N, double* c, double* s) {       // (cos, sin)(2 pi idx / N), exact at the multiples of pi / 2
    idx %= N;
    const long long i4 = 4 * idx;
    if (i4 % N == 0) {
        switch ((i4 / N) & 3) {
            case 0: *c = 1; *s = 0; return;
            case 1: *c = 0; *s = 1; return;
            case 2: *c = -1; *s = 0; return;
            default: *c = 0; *s = -1; return;
        }
    }
    const double a = 6.283185307179586476925286766559 * (double)idx / (double)N;
    *c = std::cos(a);
    *s = std::sin(a);
}
// entries: n_ent groups of (hi[64 lanes][8], lo[64 lanes][8]) bf16 = 2 KB each; value(entry, lane, j) supplied by the caller
template <class F>
static const void* b16_table(int kind, int W, int m2, int n_ent, F value) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int>, void*> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(dev, kind, W, m2, n_ent);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    std::vector<unsigned short> host((size_t)n_ent * 1024);
    for (int e = 0; e < n_ent; ++e)
        for (int ln = 0; ln < 64; ++ln)
            for (int j = 0; j < 8; ++j) {
                const float v = (float)value(e, ln, j);
                const unsigned short hi = host_bf16(v);
                const unsigned short lo = host_bf16(v - host_widen(hi));
                host[((size_t)e * 2 + 0) * 512 + ln * 8 + j] = hi;
                host[((size_t)e * 2 + 1) * 512 + ln * 8 + j] = lo;
            }
    void* d = upload_table(host.data(), host.size() * 2);
    if (!d) {
        set_error("bf16 twiddle operand table (W = %d) allocation failed: %s", W, hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    cache[key] = d;
    return d;
}

// ---------------------------------------------------------------------------------------------------------------- K1-B
// A wave owns 32-row tiles (operand tiles a0: rows 0..15, a1: rows 16..31 of the tile).
template <int NT, int MT, bool PAIRW, int XP>
__global__ __launch_bounds__(64 * B16_WAVES) void dft2d_fwd_b16_kernel(Dft2dParams p, const u32x4* __restrict__ gtab) {
    constexpr int NQ = 2 * NT;                          // operand column tiles: (cos, sin) per mode tile
    constexpr bool PAIR = MT >= 3 && PAIRW;             // +-k paired column stage (K1-HT) where its four accumulator sets fit
    constexpr int MP = PAIR ? MT / 2 + 1 : MT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int NW = p.nw;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const unsigned H8 = 8u * H;
    constexpr int xp = XP;                              // development knock-outs (timing only): 1 no row-stage MFMAs, 2 no column stage, 4 every tile reads tile 0, 16 cycle stamps

    const int nks = (W + 31) >> 5;                      // k-steps per row
    const int nch = (nks + B16_KSC - 1) / B16_KSC;      // chunks per row
    const int nkt = min(nks, B16_KSC);                  // k-steps in the table
    // LDS: [table | reduction scratch (after the main loop)] [twH]
    const size_t tab_bytes = (size_t)nkt * NQ * 2 * 1024;
    const size_t red_bytes = (size_t)(B16_WAVES / 2) * MP * NT * 8 * 64 * 4;
    u32x4* sTab = reinterpret_cast<u32x4*>(smem);                                              // [nkt][NQ][2][64]
    float* sRed = reinterpret_cast<float*>(smem);
    float2* sTwH = reinterpret_cast<float2*>(smem + std::max(tab_bytes, red_bytes));

    const int slot = wave / NW, wsub = wave - slot * NW;
    const int image = blockIdx.x * (B16_WAVES / NW) + slot;
    const bool active = image < p.n_img;
    const int nrt = (H + 31) >> 5;                      // 32-row tiles
    const int ntw = active && wsub < nrt ? (nrt - wsub + NW - 1) / NW : 0;      // this wave's tiles: wsub, wsub + NW, ...
    const bool last_tile_of_tensor = active && image == p.n_img - 1 && ntw > 0 && wsub + (ntw - 1) * NW == nrt - 1;

    // image rows through a buffer resource over [image start, end of the tensor): reads past the tensor return zero
    const unsigned short* timg = reinterpret_cast<const unsigned short*>(p.in) + (size_t)(active ? image : 0) * H * W;
    const unsigned long long span = ((unsigned long long)(p.n_img - (active ? image : 0)) * H * W * 2ull + 3ull) & ~3ull;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(timg), 0, (int)(unsigned)std::min<unsigned long long>(span, 0xfffffff0ull), 0x00020000);
    // operand of tile index ti, half hh, k-step ks (clamped: a phantom k-step of a padded last chunk re-reads the last real one)
    auto a_offset = [&](int ti, int hh, int ks) -> unsigned {
        if (xp & 4) ti = 0;
        const int row = min(32 * (wsub + min(ti, max(ntw, 1) - 1) * NW) + 16 * hh + r16, H - 1);
        return (unsigned)((row * W + 32 * min(ks, nks - 1) + 8 * kk) * 2);
    };
    // chunk position P = (tile index, reversed chunk index) in this wave's order; the ring holds the chunk being multiplied while the
    // next one is requested slot by slot
    u32x4 ring[B16_KSC][2];
    const unsigned last_pixel = 0xffffu & (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (unsigned)((H * W - 1) * 2), 0, 0);
    int lti = 0, lcc = 0;                               // position of the next chunk to request
#pragma unroll
    for (int s = 0; s < B16_KSC; ++s)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            ring[s][hh] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a_offset(0, hh, B16_KSC * (nch - 1) + s), 0, 0);
    if (++lcc == nch) { lcc = 0; ++lti; }

    // ---- tables (copied while the first loads are on their way)
    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];
    for (int e = tid; e < nkt * NQ * 2 * 64; e += nthreads) sTab[e] = gtab[e];
    __syncthreads();

    // column-stage accumulators exactly as in K1-HT (dft2d_fwd_ht_kernel.h): !PAIR: Xr / Xi = Re / Im of the spectrum rows;
    // PAIR: Xr = Re C, Xi = Re S, Yr = -Im C, Yi = -Im S with C_k = sum_h cos(theta_k h) T[h], S_k = sum_h sin(theta_k h) T[h]
    f32x4 Xr[MP][NT], Xi[MP][NT], Yr[PAIR ? MP : 1][NT], Yi[PAIR ? MP : 1][NT];
#pragma unroll
    for (int mt = 0; mt < MP; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Xr[mt][t] = f32x4{0, 0, 0, 0}; Xi[mt][t] = f32x4{0, 0, 0, 0};
            if (PAIR || mt == 0) { Yr[PAIR ? mt : 0][t] = f32x4{0, 0, 0, 0}; Yi[PAIR ? mt : 0][t] = f32x4{0, 0, 0, 0}; }
        }
    // rotation between chunks: (Tr - i Tn) *= (cr - i sr), (cr, sr) = (cos, sin)(2 pi l KC / W), l = 16 t + r16
    float2 rho[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        rho[t] = p.twW[(unsigned)(((unsigned long long)(32 * B16_KSC) * (unsigned)(16 * t + r16)) % (unsigned)W)];

    const u32x4* tabL = sTab + lane;
    // twiddle operands of the NEXT k-step are requested behind the MFMAs that read the current ones (hi behind hi, lo behind lo)
    u32x4 bh[NQ], bl[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { bh[q] = tabL[(q * 2 + 0) * 64]; bl[q] = tabL[(q * 2 + 1) * 64]; }
    f32x4 T[2][NQ];                     // row-stage accumulators of the tile's two halves: T[hh][2 t] = Re T, T[hh][2 t + 1] = -Im T = sum x sin
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int q = 0; q < NQ; ++q) T[hh][q] = f32x4{0, 0, 0, 0};
    int ti = 0, cc = 0;
    const int totalP = ntw * nch;
    unsigned long long t_prev = b16_clock(), t_rot = 0, t_rowst = 0, t_colst = 0, t_wait = 0;
    const unsigned long long t_begin = t_prev, r_begin = __builtin_amdgcn_s_memrealtime();
    for (int P = 0; P < totalP; ++P) {
        const int c = nch - 1 - cc;                     // chunks from the last to the first
        if (cc == 0) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int q = 0; q < NQ; ++q) T[hh][q] = f32x4{0, 0, 0, 0};
        } else {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float tr = T[hh][2 * t][r], tn = T[hh][2 * t + 1][r];
                        T[hh][2 * t][r] = rho[t].x * tr - rho[t].y * tn;
                        T[hh][2 * t + 1][r] = rho[t].x * tn + rho[t].y * tr;
                    }
        }
        const int lc = nch - 1 - lcc;                   // chunk of the position being requested
        B16_STAMP(t_rot);
#pragma unroll
        for (int s = 0; s < B16_KSC; ++s) {
            const int ks = B16_KSC * c + s;
            u32x4& a0 = ring[s][0];         // (the slot is re-requested BEHIND the MFMAs that read it: no copies of the operands)
            u32x4& a1 = ring[s][1];
            const int sn = (s + 1 < nkt) ? s + 1 : 0;       // (the table of a short row holds fewer than 8 k-steps)
            if (32 * ks + 32 > W) {         // ragged end of the row (and phantom k-steps): columns >= W count as zero
                const int nv = W - 32 * ks - 8 * kk;        // valid elements of this lane's eight
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned m = nv >= 2 * d + 2 ? 0xffffffffu : (nv == 2 * d + 1 ? 0x0000ffffu : 0u);
                    a0[d] &= m; a1[d] &= m;
                }
                // the LAST pixel of the tensor, when its row starts 2-byte aligned (odd row length): it is the low half of a
                // dword that straddles the end of the buffer, and the range check returns zero for such a dword: it was fetched
                // as a 2-byte access in the prologue (a load HERE, under a branch, makes the compiler drain the whole ring with
                // s_waitcnt vmcnt(0) at every k-step: measured 1 500 instead of 300 cycles per k-step)
                if ((W & 1) && last_tile_of_tensor && ks == nks - 1 && ti == ntw - 1) {
                    const int je = nv - 1;
                    const int row0 = 32 * (wsub + ti * NW) + r16;
                    if (nv >= 1 && nv <= 8 && (je & 1) == 0 && (row0 == H - 1 || row0 + 16 == H - 1)) {
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            if (je == 2 * d) { if (row0 == H - 1) a0[d] = last_pixel; else a1[d] = last_pixel; }
                    }
                }
            }
            // (phantom k-steps of a padded last chunk multiply an all-zero operand: no branch around the MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(xp & 1)) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) { T[0][q] = mfma_b16(a0, bh[q], T[0][q]); T[1][q] = mfma_b16(a1, bh[q], T[1][q]); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) bh[q] = tabL[((sn * NQ + q) * 2 + 0) * 64];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(xp & 1)) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) { T[0][q] = mfma_b16(a0, bl[q], T[0][q]); T[1][q] = mfma_b16(a1, bl[q], T[1][q]); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) bl[q] = tabL[((sn * NQ + q) * 2 + 1) * 64];
            ring[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a_offset(lti, 0, B16_KSC * lc + s), 0, 0);
            ring[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a_offset(lti, 1, B16_KSC * lc + s), 0, 0);
        }
        if (++lcc == nch) { lcc = 0; ++lti; }
        B16_STAMP(t_rowst);
        if (++cc == nch) {
            cc = 0;
            const int rt = wsub + ti * NW;
            ++ti;
            if constexpr (!(xp & 2)) {
                // ---- column stage, one 16-row half at a time: X[j][l] += exp(-i theta(j, h)) T[h][l] (K1-HT's stage B)
                // (the row frequencies are recomputed per tile: registers that would otherwise live through the row stage)
                int Kj[MP];
                bool jvalid[MP];
#pragma unroll
                for (int mt = 0; mt < MP; ++mt) {
                    const int j = 16 * mt + r16;
                    if constexpr (PAIR) {
                        jvalid[mt] = j <= m1;
                        Kj[mt] = jvalid[mt] ? j : 0;
                    } else {
                        jvalid[mt] = j < 2 * m1;
                        Kj[mt] = jvalid[mt] ? corner_freq(j, m1, H) : 0;
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int h0 = 32 * rt + 16 * hh + 4 * kk;
                    unsigned idxB[MP];
                    float2 twB[MP];
#pragma unroll
                    for (int mt = 0; mt < MP; ++mt) {
                        const unsigned i0 = 8u * (((unsigned)Kj[mt] * (unsigned)h0) % (unsigned)H);
                        twB[mt] = lds_tw(sTwH, i0);
                        idxB[mt] = wrap_add(i0, 8u * (unsigned)Kj[mt], H8);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const bool hvalid = (h0 + s) < H;
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
                                    Xr[mt][t] = mfma16(ac, T[hh][2 * t][s], Xr[mt][t]);
                                    Yr[mt][t] = mfma16(ac, T[hh][2 * t + 1][s], Yr[mt][t]);
                                    Xi[mt][t] = mfma16(as, T[hh][2 * t][s], Xi[mt][t]);
                                    Yi[mt][t] = mfma16(as, T[hh][2 * t + 1][s], Yi[mt][t]);
                                }
                            } else {
                                const float ans = v ? -twB[mt].y : 0.f;
                                const float anc = -ac;
#pragma unroll
                                for (int t = 0; t < NT; ++t) {
                                    Xr[mt][t] = mfma16(ac, T[hh][2 * t][s], Xr[mt][t]);
                                    Xi[mt][t] = mfma16(anc, T[hh][2 * t + 1][s], Xi[mt][t]);
                                    Xr[mt][t] = mfma16(ans, T[hh][2 * t + 1][s], Xr[mt][t]);
                                    Xi[mt][t] = mfma16(ans, T[hh][2 * t][s], Xi[mt][t]);
                                }
                            }
                        }
#pragma unroll
                        for (int mt = 0; mt < MP; ++mt) twB[mt] = twBn[mt];
                    }
                }
            }
            B16_STAMP(t_colst);
        }
    }
    if ((xp & 16) && p.rowfreq && lane == 0) {
        unsigned long long* outp = reinterpret_cast<unsigned long long*>(const_cast<int*>(p.rowfreq)) + ((size_t)blockIdx.x * B16_WAVES + wave) * 4;
        outp[0] = t_rot; outp[1] = t_rowst; outp[2] = t_colst;
        outp[3] = (b16_clock() - t_begin) * 1000ull / (__builtin_amdgcn_s_memrealtime() - r_begin + 1);       // shader cycles per 10 us of the 100 MHz clock: MHz / 10
    }

    // ---- several waves per image: deterministic tree reduction through LDS (the table is no longer needed)
    if (NW > 1) __syncthreads();
    const int rbuf = MP * NT * 8 * 64;                          // floats per receiver buffer
    auto reduce_pair = [&](f32x4 (*A)[NT], f32x4 (*Bv)[NT], int stride) {
        float* mine = sRed + (size_t)(slot * (NW / 2) + (wsub < stride ? wsub : wsub - stride)) * rbuf;
        if (wsub >= stride && wsub < 2 * stride) {
#pragma unroll
            for (int mt = 0; mt < MP; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mine[((mt * NT + t) * 8 + r) * 64 + lane] = A[mt][t][r];
                        mine[((mt * NT + t) * 8 + 4 + r) * 64 + lane] = Bv[mt][t][r];
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
                        A[mt][t][r] += mine[((mt * NT + t) * 8 + r) * 64 + lane];
                        Bv[mt][t][r] += mine[((mt * NT + t) * 8 + 4 + r) * 64 + lane];
                    }
        }
        __syncthreads();
    };
    for (int stride = B16_WAVES / 2; stride >= 1; stride >>= 1) {
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

// waves per image for n_img images on this device: fill every CU first, then several images per workgroup
static void b16_geometry(int n_img, int nrt, int waves, int* nw, int* g) {
    const int cus = b16_cu_count();
    int G = 1;
    while (G < waves && (long long)n_img >= (long long)cus * 2 * G) G *= 2;
    int NW = waves / G;
    while (NW > 1 && NW > nrt) NW >>= 1;
    *nw = NW;
    *g = waves / NW;
}

static size_t fwd_b16_lds(const Dft2dParams& p, int NT, int MP) {
    const int nks = (p.W + 31) / 32, nkt = std::min(nks, B16_KSC);
    const size_t tab = (size_t)nkt * 2 * NT * 2 * 1024, red = (size_t)(B16_WAVES / 2) * MP * NT * 8 * 64 * 4;
    return std::max(tab, red) + (size_t)p.H * 8;
}

template <int NT, int MT, bool PAIRW, int XP>
static int launch_fwd_b16_v(Dft2dParams p, hipStream_t s) {
    constexpr bool PAIR = MT >= 3 && PAIRW;
    constexpr int MP = PAIR ? MT / 2 + 1 : MT;
    constexpr int NQ = 2 * NT;
    const int W = p.W, m2 = p.m2;
    const int nks = (W + 31) / 32, nkt = std::min(nks, B16_KSC);
    // entry e = (k-step ks, column tile q): lane (n = ln & 15, g = ln >> 4), element j: column wc = 32 ks + 8 g + j of the chunk,
    // mode l = 16 (q >> 1) + n; q even: cos, q odd: sin (the kernel accumulates Re T and -Im T)
    const void* tab = b16_table(0, W, m2 * 64 + NT, nkt * NQ, [=](int e, int ln, int j) -> double {
        const int q = e % NQ, ks = e / NQ;
        const int l = 16 * (q >> 1) + (ln & 15);
        if (l >= m2) return 0.0;
        double c, sn;
        host_cossin((long long)(32 * ks + 8 * (ln >> 4) + j) * l, W, &c, &sn);
        return (q & 1) ? sn : c;
    });
    if (!tab) return -6;
    const size_t lds = fwd_b16_lds(p, NT, MP);
    if (lds > 160 * 1024) { set_error("dft2d_fwd_b16: %zu bytes of LDS", lds); return -3; }       // (the caller falls back to the f32-MFMA form)
    auto k = dft2d_fwd_b16_kernel<NT, MT, PAIRW, XP>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_slot)) { set_error("dft2d_fwd_b16: cannot raise dynamic LDS to %zu", lds); return -4; }
    int nw, g;
    b16_geometry(p.n_img, (p.H + 31) / 32, B16_WAVES, &nw, &g);
    p.nw = nw;
    if (XP & 16) p.rowfreq = reinterpret_cast<const int*>((uintptr_t)strtoull(getenv("UNO_B16_STAMPS") ? getenv("UNO_B16_STAMPS") : "0", nullptr, 0));
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_fwd_b16_kernel<%d, %d, %d>", NT, MT, (int)PAIRW);
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * 2.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3((p.n_img + g - 1) / g), dim3(64 * B16_WAVES), lds, s, p, reinterpret_cast<const u32x4*>(tab));
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_fwd_b16 launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

template <int NT, int MT>
static int launch_fwd_b16_t(const Dft2dParams& p, hipStream_t s) {
#ifdef UNO_B16_DEV          // development build (tools/dev/mkvariant.py): knock-out / stamp instantiations selected by the environment
    if constexpr (NT == 2 && MT == 4) {
        static const int xp = b16_exp("UNO_B16_FWD_EXP", 0);
        switch (xp) {
            case 1: return launch_fwd_b16_v<NT, MT, (NT * MT < 6), 1>(p, s);
            case 2: return launch_fwd_b16_v<NT, MT, (NT * MT < 6), 2>(p, s);
            case 4: return launch_fwd_b16_v<NT, MT, (NT * MT < 6), 4>(p, s);
            case 6: return launch_fwd_b16_v<NT, MT, (NT * MT < 6), 6>(p, s);
            case 16: return launch_fwd_b16_v<NT, MT, (NT * MT < 6), 16>(p, s);
            default: break;
        }
    }
#endif
    return launch_fwd_b16_v<NT, MT, (NT * MT < 6), 0>(p, s);
}

// ---------------------------------------------------------------------------------------------------------------- K3-B
// NC = 16 NCT output columns per chunk (one twiddle table; NCT is a template parameter); a wave owns 32-row tiles (two 16-row operand tiles); staging: 32 rows x
// NC bf16 per wave.  The +-k pair operands of the column stage (72 registers at 32 x 32 modes) live in LDS per image, not in
// registers: they are read once per tile.
template <int NT, int NCT, int SCT, int XP>
__global__ __launch_bounds__(64 * B16_WAVES) void dft2d_inv_b16_kernel(Dft2dParams p, const u32x4* __restrict__ gtab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int NW = p.nw;
    const int G = B16_WAVES / NW;               // images per workgroup
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const unsigned H8 = 8u * H;
    constexpr int NC = 16 * NCT;                                // columns per chunk: compile-time, so that the group and flush loops unroll
    const int nch = (W + NC - 1) / NC;
    constexpr int SC = 16 * SCT;                                // columns staged and flushed at a time (NCT / SCT flushes per chunk)
    constexpr int srow = SC / 2 + 4;                            // staging row pitch in dwords: SC bf16 + 16 bytes (bank skew)
    const int ksk = (m1 + 4) >> 2;                              // k-steps of the paired column stage: k = 0 .. m1 in fours (<= 2 JT + 1)
    u32x4* sTab = reinterpret_cast<u32x4*>(smem);                                              // [NCT][NT][2][64]
    unsigned* sStage = reinterpret_cast<unsigned*>(smem + (size_t)NCT * NT * 2 * 1024);       // [waves][32][srow]
    float2* sRho = reinterpret_cast<float2*>(sStage + (size_t)B16_WAVES * 32 * srow);         // [nch][16 NT]
    float2* sTwH = sRho + (size_t)nch * 16 * NT;
    float* sPM = reinterpret_cast<float*>(sTwH + ((H + 1) & ~1));                              // [G][NT][ksk][4][64]: Pr, Pi, Mr, Mi operands

    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];
    for (int e = tid; e < nch * 16 * NT; e += nthreads) {
        const int c = e / (16 * NT), l = e - c * 16 * NT;
        sRho[e] = p.twW[(unsigned)(((unsigned long long)c * (unsigned)NC * (unsigned)l) % (unsigned)W)];
    }
    for (int e = tid; e < NCT * NT * 2 * 64; e += nthreads) sTab[e] = gtab[e];
    // column-stage A operands: the corner rows come in +-k pairs (K3, dft2d_inv_kernel.h): P_k = O[+k] + O[-k], M_k = O[+k] - O[-k];
    // operand lane (rho = ln & 15 -> mode 16 t + 4 (rho & 3) + (rho >> 2), k-slot ln >> 4 -> k = 4 ks + (ln >> 4))
    for (int e = tid; e < G * NT * ksk * 64; e += nthreads) {
        const int ln = e & 63, ks = (e >> 6) % ksk, t = ((e >> 6) / ksk) % NT, gi = (e >> 6) / (ksk * NT);
        const int img_i = blockIdx.x * G + gi;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (img_i < p.n_img) {
            const float2* O = reinterpret_cast<const float2*>(p.in) + spectrum_index(p, img_i) * 2 * m1 * m2;
            const int l = 16 * t + 4 * (ln & 3) + ((ln & 15) >> 2);
            const int k = 4 * ks + (ln >> 4);
            const float cs = p.scale * ((p.herm && l < m2) ? herm_weight(l, W) : 1.0f);
            float2 vp = make_float2(0.f, 0.f), vm = make_float2(0.f, 0.f);
            if (l < m2 && k < m1 && !(p.mask && !row_survives(k, m1, H))) vp = O[(size_t)k * m2 + l];
            if (l < m2 && k >= 1 && k <= m1) vm = O[(size_t)(2 * m1 - k) * m2 + l];
            v = make_float4((vp.x + vm.x) * cs, (vp.y + vm.y) * cs, (vp.x - vm.x) * cs, (vp.y - vm.y) * cs);
        }
        float* dst = sPM + ((size_t)((gi * NT + t) * ksk + ks) * 4) * 64 + ln;
        dst[0] = v.x; dst[64] = v.y; dst[128] = v.z; dst[192] = v.w;
    }
    __syncthreads();

    const int slot = wave / NW, wsub = wave - slot * NW;
    const int image = blockIdx.x * G + slot;
    if (image >= p.n_img) return;               // no barrier below

    unsigned short* img = reinterpret_cast<unsigned short*>(p.out) + (size_t)image * H * W;
    const int nrt = (H + 31) >> 5;              // 32-row tiles
    constexpr int xp = XP;                      // development knock-outs (timing only): 1 no row-stage MFMAs, 2 no global stores, 4 no staging, 8 no rotation, 16 cycle stamps
    unsigned* stg = sStage + (size_t)wave * 32 * srow;
    const u32x4* tabL = sTab + lane;
    const float* pmL = sPM + (size_t)slot * NT * ksk * 256 + lane;
    const bool w8 = (W & 7) == 0;               // rows 16-byte aligned: one 16-byte store per lane
    constexpr int lpr = SC >> 3;                // flush: lanes per row segment (a lane owns 8 columns = 16 bytes) ...
    constexpr int RP = 64 / lpr;                // ... rows per pass
    const int frow = lane / lpr, fcol = lane - frow * lpr;

    unsigned long long t_prev = b16_clock(), t_col = 0, t_split = 0, t_row = 0, t_flush = 0;
    for (int rt = wsub; rt < nrt; rt += NW) {
        // ---- column stage of both halves: U^T[mode][h], lane (g = kk, h = r16), register r of tile t <-> mode 16 t + 4 r + g
        f32x4 Ur[2][NT], Ui[2][NT];
        {
            unsigned a4[2], aj[2];
            float2 twb[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const unsigned hB = (unsigned)min(32 * rt + 16 * hh + r16, H - 1);
                a4[hh] = 8u * ((4u * hB) % (unsigned)H);
                aj[hh] = 8u * (((unsigned)kk * hB) % (unsigned)H);
                twb[hh] = lds_tw(sTwH, aj[hh]);
#pragma unroll
                for (int t = 0; t < NT; ++t) { Ur[hh][t] = f32x4{0, 0, 0, 0}; Ui[hh][t] = f32x4{0, 0, 0, 0}; }
            }
            for (int ks = 0; ks < ksk; ++ks) {
                float2 twn[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    aj[hh] = wrap_add(aj[hh], a4[hh], H8);
                    twn[hh] = lds_tw(sTwH, aj[hh]);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float* pm = pmL + (size_t)(t * ksk + ks) * 256;
                    const float pr = pm[0], pi = pm[64], mr = pm[128], mi = pm[192];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const float ns = -twb[hh].y;
                        Ur[hh][t] = mfma16(pr, twb[hh].x, Ur[hh][t]);
                        Ui[hh][t] = mfma16(pi, twb[hh].x, Ui[hh][t]);
                        Ur[hh][t] = mfma16(mi, ns, Ur[hh][t]);
                        Ui[hh][t] = mfma16(mr, twb[hh].y, Ui[hh][t]);
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) twb[hh] = twn[hh];
            }
        }

        B16_STAMP(t_col);
        for (int c = 0; c < nch; ++c) {
            // ---- U rotated to the chunk, split into hi + lo bf16 operands: k-step ks holds values 8 ks .. 8 ks + 7 of the list
            // [Ur of tile 0 (r = 0..3), Ur of tile 1, ..., Ui of tile 0, ...]
            u32x4 Uh[2][NT], Ul[2][NT];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float V[8 * NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float ur = Ur[hh][t][r], ui = Ui[hh][t][r];
                        if (c == 0 || (xp & 8)) {
                            V[4 * t + r] = ur; V[4 * NT + 4 * t + r] = ui;
                        } else {
                            const float2 rho = sRho[c * 16 * NT + 16 * t + 4 * r + kk];
                            V[4 * t + r] = ur * rho.x - ui * rho.y;
                            V[4 * NT + 4 * t + r] = ur * rho.y + ui * rho.x;
                        }
                    }
#pragma unroll
                for (int ks = 0; ks < NT; ++ks)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        unsigned h, l;
                        split2(V[8 * ks + 2 * jj], V[8 * ks + 2 * jj + 1], h, l);
                        Uh[hh][ks][jj] = h; Ul[hh][ks][jj] = l;
                    }
            }
            B16_STAMP(t_split);
            // ---- row stage: two column tiles x two halves at a time (four independent accumulator chains per twiddle read)
            const int col0 = c * NC;
            const int nmt = min(NCT, (W - col0 + 15) >> 4);
#pragma unroll
            for (int sub = 0; sub < NCT / SCT; ++sub) {
            if (sub * SCT >= nmt) break;
#pragma unroll
            for (int m0 = sub * SCT; m0 < (sub + 1) * SCT; m0 += 2) {
                if (m0 >= nmt) break;
                f32x4 D[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { D[u][0] = f32x4{0, 0, 0, 0}; D[u][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll
                for (int ks = 0; ks < NT; ++ks) {
                    u32x4 Ah[2], Al[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int mt = min(m0 + u, NCT - 1);
                        Ah[u] = tabL[((mt * NT + ks) * 2 + 0) * 64];
                        Al[u] = tabL[((mt * NT + ks) * 2 + 1) * 64];
                    }
                    if (!(xp & 1)) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) { D[u][0] = mfma_b16(Ah[u], Uh[0][ks], D[u][0]); D[u][1] = mfma_b16(Ah[u], Uh[1][ks], D[u][1]); }
#pragma unroll
                        for (int u = 0; u < 2; ++u) { D[u][0] = mfma_b16(Al[u], Uh[0][ks], D[u][0]); D[u][1] = mfma_b16(Al[u], Uh[1][ks], D[u][1]); }
#pragma unroll
                        for (int u = 0; u < 2; ++u) { D[u][0] = mfma_b16(Ah[u], Ul[0][ks], D[u][0]); D[u][1] = mfma_b16(Ah[u], Ul[1][ks], D[u][1]); }
                    } else {
#pragma unroll
                        for (int u = 0; u < 2; ++u) D[u][0][0] += __uint_as_float(Ah[u][0] ^ Al[u][1] ^ Uh[0][ks][2] ^ Ul[1][ks][3]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (m0 + u < nmt && !(xp & 4)) {
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            unsigned* dst = stg + (16 * hh + r16) * srow + 8 * (m0 + u - sub * SCT) + 2 * kk;      // columns 16 mt + 4 kk .. + 3 of row 16 hh + r16
                            *reinterpret_cast<uint2*>(dst) = make_uint2(bf16_pack2(D[u][hh][0], D[u][hh][1]), bf16_pack2(D[u][hh][2], D[u][hh][3]));
                        }
                    }
                }
            }
            B16_STAMP(t_row);
            // ---- the staged columns leave as row segments: pass q covers rows q * RP .. + RP - 1, a lane owns 8 columns (16 bytes)
            const int col = col0 + sub * SC + 8 * fcol;
            u32x4 fv[32 / RP];
#pragma unroll
            for (int q = 0; q < 32 / RP; ++q) fv[q] = *reinterpret_cast<const u32x4*>(stg + (q * RP + frow) * srow + 4 * fcol);
#pragma unroll
            for (int q = 0; q < 32 / RP; ++q) {
                const int row = q * RP + frow;
                const int h = 32 * rt + row;
                const u32x4 v = fv[q];
                if (h < H && col < W && !(xp & 2)) {
                    unsigned short* dst = img + (size_t)h * W + col;
                    if (col + 8 <= W) {
                        if (w8) {
                            *reinterpret_cast<u32x4*>(dst) = v;
                        } else {
                            *reinterpret_cast<h8u*>(dst) = h8u{{v[0], v[1], v[2], v[3]}};
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (col + e < W) dst[e] = (unsigned short)((v[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                    }
                }
            }
            B16_STAMP(t_flush);
            }
        }
    }
    if ((xp & 16) && p.rowfreq && lane == 0) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<int*>(p.rowfreq)) + ((size_t)blockIdx.x * B16_WAVES + wave) * 4;
        out[0] = t_col; out[1] = t_split; out[2] = t_row; out[3] = t_flush;
    }
}

static size_t inv_b16_lds(const Dft2dParams& p, int NT, int NCT, int SCT, int G) {
    const int NC = 16 * NCT, nch = (p.W + NC - 1) / NC, ksk = (p.m1 + 4) >> 2;
    return (size_t)NCT * NT * 2 * 1024 + (size_t)B16_WAVES * 32 * (16 * SCT / 2 + 4) * 4 + (size_t)nch * 16 * NT * 8 + (size_t)((p.H + 1) & ~1) * 8 +
           (size_t)G * NT * ksk * 1024;
}

template <int NT, int NCT, int SCT, int XP>
static int launch_inv_b16_v(Dft2dParams p, hipStream_t s) {
    int nw, g;
    b16_geometry(p.n_img, (p.H + 31) / 32, B16_WAVES, &nw, &g);
    p.nw = nw;
    const size_t lds = inv_b16_lds(p, NT, NCT, SCT, g);
    if (lds > 160 * 1024) { set_error("dft2d_inv_b16: %zu bytes of LDS", lds); return -3; }
    const int W = p.W, m2 = p.m2;
    // entry e = (column tile mt, k-step ks): lane (i = ln & 15 -> column wc = 16 mt + i of the chunk, g = ln >> 4), element j: value
    // v = 8 ks + j of the lane's list [Ur of tile 0 (r = 0..3), Ur of tile 1, ..., Ui of tile 0, ...]; value (t, r) is mode 16 t + 4 r + g
    const void* tab = b16_table(1, W, m2 * 64 + NT, NCT * NT, [=](int e, int ln, int j) -> double {
        const int ks = e % NT, mt = e / NT;
        const int vi = 8 * ks + j;
        const bool im = vi >= 4 * NT;
        const int vv = im ? vi - 4 * NT : vi;
        const int l = 16 * (vv >> 2) + 4 * (vv & 3) + (ln >> 4);
        if (l >= m2) return 0.0;
        double c, sn;
        host_cossin((long long)(16 * mt + (ln & 15)) * l, W, &c, &sn);
        return im ? -sn : c;
    });
    if (!tab) return -6;
    auto k = dft2d_inv_b16_kernel<NT, NCT, SCT, XP>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_slot)) { set_error("dft2d_inv_b16: cannot raise dynamic LDS to %zu", lds); return -4; }
    if (XP & 16) p.rowfreq = reinterpret_cast<const int*>((uintptr_t)strtoull(getenv("UNO_B16_STAMPS") ? getenv("UNO_B16_STAMPS") : "0", nullptr, 0));
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_inv_b16_kernel<%d, %d, %d>", NT, NCT, SCT);
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * 2.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3((p.n_img + g - 1) / g), dim3(64 * B16_WAVES), lds, s, p, reinterpret_cast<const u32x4*>(tab));
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_inv_b16 launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

// chunk = 256 columns (one rotation + split of U per 256 columns), staged and flushed 128 at a time, where the 64 KB table, the
// staging and the column-stage operands fit the LDS together; else 128 / 128; short rows 64 / 64
template <int NT>
static int launch_inv_b16_t(const Dft2dParams& p, hipStream_t s) {
    int nw, g;
    b16_geometry(p.n_img, (p.H + 31) / 32, B16_WAVES, &nw, &g);
    const size_t limit = 160 * 1024;
    const bool c256 = p.W > 256 && inv_b16_lds(p, NT, 16, 8, g) <= limit;
    const bool c128 = p.W > 128 && inv_b16_lds(p, NT, 8, 8, g) <= limit;
#ifdef UNO_B16_DEV
    if (NT == 2 && c128) {
        static const int xp = b16_exp("UNO_B16_INV_EXP", 0);
        switch (xp) {
            case 1: return launch_inv_b16_v<NT, 8, 8, 1>(p, s);
            case 2: return launch_inv_b16_v<NT, 8, 8, 2>(p, s);
            case 4: return launch_inv_b16_v<NT, 8, 8, 4>(p, s);
            case 8: return launch_inv_b16_v<NT, 8, 8, 8>(p, s);
            case 16: return c256 ? launch_inv_b16_v<NT, 16, 8, 16>(p, s) : launch_inv_b16_v<NT, 8, 8, 16>(p, s);
            case 32: return launch_inv_b16_v<NT, 8, 8, 0>(p, s);          // A/B: 128-column chunks where 256 would fit
            default: break;
        }
    }
#endif
    if (c256) return launch_inv_b16_v<NT, 16, 8, 0>(p, s);
    return c128 ? launch_inv_b16_v<NT, 8, 8, 0>(p, s) : launch_inv_b16_v<NT, 4, 4, 0>(p, s);
}

// ---------------------------------------------------------------------------------------------------------------- dispatch
// bf16 images, corner rule (no frequency tables), the compiled mode range, rows long enough for a k-step to be mostly data
bool dft2d_b16_applies(const Dft2dParams& p) {
    if (!p.bf16 || p.rowfreq) return false;
#ifdef UNO_B16_DEV
    static const int off = b16_exp("UNO_B16_OFF", 0);           // development build only: A/B against the f32-MFMA forms
    if (off) return false;
#endif
    if (p.m1 > 40 || p.m2 > 32 || p.W < 64 || p.H < 16) return false;               // (48 column modes: three accumulator tiles per half spill)
    if (((p.m2 + 15) / 16) * ((2 * p.m1 + 15) / 16) > 8) return false;              // those instantiations spill: the f32-MFMA forms keep them
    if ((unsigned long long)p.H * p.W * 2ull >= 0x7fffffffull) return false;       // 32-bit byte offsets inside an image
    return true;
}

int launch_dft2d_fwd_b16(const Dft2dParams& p, hipStream_t s) {
    const int NT = (p.m2 + 15) / 16, MT = (2 * p.m1 + 15) / 16;
#define UNO_CASE(nt, mt) if (NT == nt && MT == mt) return launch_fwd_b16_t<nt, mt>(p, s);
    UNO_CASE(1, 1) UNO_CASE(1, 2) UNO_CASE(1, 3) UNO_CASE(1, 4) UNO_CASE(1, 5)
    UNO_CASE(2, 1) UNO_CASE(2, 2) UNO_CASE(2, 3) UNO_CASE(2, 4)
#undef UNO_CASE
    set_error("dft2d_fwd_b16: modes (%d, %d) exceed the compiled range", p.m1, p.m2);
    return -2;
}

int launch_dft2d_inv_b16(const Dft2dParams& p, hipStream_t s) {
    const int NT = (p.m2 + 15) / 16;
    if (NT == 1) return launch_inv_b16_t<1>(p, s);
    if (NT == 2) return launch_inv_b16_t<2>(p, s);
    set_error("dft2d_inv_b16: modes (%d, %d) exceed the compiled range", p.m1, p.m2);
    return -2;
}

}  // namespace uno
