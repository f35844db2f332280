// K7 - separable banded resampling of images:  out = A . in . B^T  per image,
//   A (Ho x H) and B (Wo x W) banded matrices given as (first source index, taps) per output index.
//
// This is the resampling half of pointwise_op_2D (reference integral_operators.py:240-242:
// F.interpolate(mode="bicubic", align_corners=True, antialias=True)) and, with the transposed band tables, its
// adjoint.  The band tables hold exactly the weights the reference's CPU op applies (they are read off that op
// once per (in, out) size pair on the host, uno_amd/resample.py), so the forward result matches the reference
// to float32 rounding; what this kernel replaces is the GPU implementation of the op, whose backward
// (upsample_gen2d_aa_backward_out_frame) took 196 ms of a 286 ms training step.
//
// Memory-bound stencil, two passes (rows then columns, or columns then rows - whichever makes the
// intermediate smaller); every load and store is unit-stride along the image row.
#include "uno_common.h"
#include <algorithm>
#include <cstdio>
#include <type_traits>

namespace uno {

// out[n][i][q] = sum_t wt[i][t] * in[n][start[i] + t][q]      (rows of length W)
// TI / TO: element types of the source / destination (float | unsigned short = bfloat16 bits); the two-pass form keeps its
// intermediate in f32
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void resample_rows_kernel(const TI* __restrict__ in, TO* __restrict__ out,
                                                            const int* __restrict__ start, const float* __restrict__ wt,
                                                            int K, int H, int Ho, int W, int rows_per_block, int accumulate) {
    const int n = blockIdx.z;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= W) return;
    const TI* src = in + (size_t)n * H * W + q;
    TO* dst = out + (size_t)n * Ho * W + q;
    const int i0 = blockIdx.y * rows_per_block;
    const int i1 = min(i0 + rows_per_block, Ho);
    for (int i = i0; i < i1; ++i) {
        const int s = start[i];
        const float* w = wt + (size_t)i * K;
        float acc = 0.f;
        for (int t = 0; t < K; ++t) {
            const int p = min(s + t, H - 1);            // taps beyond the band carry weight 0
            acc = fmaf(w[t], io_widen(src[(size_t)p * W]), acc);
        }
        io_store1(dst + (size_t)i * W, accumulate ? io_widen(dst[(size_t)i * W]) + acc : acc);
    }
}

// out[n][r][j] = sum_t wt[j][t] * in[n][r][start[j] + t]
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void resample_cols_kernel(const TI* __restrict__ in, TO* __restrict__ out,
                                                            const int* __restrict__ start, const float* __restrict__ wt,
                                                            int K, int R, int W, int Wo, int rows_per_block, int accumulate) {
    const int n = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Wo) return;
    const int s = start[j];
    float w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = t < K ? wt[(size_t)j * K + t] : 0.f;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(r0 + rows_per_block, R);
    for (int r = r0; r < r1; ++r) {
        const TI* src = in + ((size_t)n * R + r) * W;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < K) acc = fmaf(w[t], io_widen(src[min(s + t, W - 1)]), acc);
        TO* o = out + ((size_t)n * R + r) * Wo + j;
        io_store1(o, accumulate ? io_widen(*o) + acc : acc);
    }
}

// any band width: the taps in a run-time loop, weights read per tap (strong down-sampling - 90 -> 8 columns has 43 taps - and the
// adjoints of strong up-sampling; the U-NO models stay below 11 taps and never come here)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void resample_cols_generic_kernel(const TI* __restrict__ in, TO* __restrict__ out,
                                                                    const int* __restrict__ start, const float* __restrict__ wt,
                                                                    int K, int R, int W, int Wo, int rows_per_block, int accumulate) {
    const int n = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Wo) return;
    const int s = start[j];
    const float* w = wt + (size_t)j * K;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(r0 + rows_per_block, R);
    for (int r = r0; r < r1; ++r) {
        const TI* src = in + ((size_t)n * R + r) * W;
        float acc = 0.f;
        for (int t = 0; t < K; ++t) acc = fmaf(w[t], io_widen(src[min(s + t, W - 1)]), acc);
        TO* o = out + ((size_t)n * R + r) * Wo + j;
        io_store1(o, accumulate ? io_widen(*o) + acc : acc);
    }
}

// Fused form: one workgroup = (image, tile of TR = 16 output rows).
//   phase 1 (rows): every thread owns one image column.  It streams the NP input rows the tile depends on - each
//     element loaded exactly once, unit-stride across the wave, all loads independent - and accumulates the 16
//     outputs as a DENSE 16 x NP tile of the banded row operator whose weights are wave-uniform (scalar registers):
//     16 v_fmac per loaded element instead of a gather per tap.  Result V[16][W] goes to LDS.
//   phase 2 (columns): banded column operator from LDS, TR x Wo outputs written unit-stride.
// The image is read once (plus the band overlap of neighbouring tiles, an L2 hit) and the result written once.
constexpr int RS_TR = 16;
// phase-1 prefetch depth (k-steps of four input rows in flight per lane).  Measured 1 .. 8 on one box (tools/rsbench.py, six Darcy
// shapes + the accumulating calls): 2 is best, 3 (rounds 2-3) 2-4 % behind, 6 and 8 are 5-25 % SLOWER - more loads in flight per
// wave hurt this kernel (Darcy step 13.95 -> 13.84 ms with 2).  Re-measured at the end of round 6 under the 64-register cap (3 and 4 still fit
// it: 61 / 64 registers, 8 waves per SIMD): 446 -> 223 270 / 269 / 302 us, the accumulating 223 -> 446 389 / 435 / 491 us for 2 / 3 / 4.
#ifndef UNO_RS_PD
#define UNO_RS_PD 2
#endif
constexpr int RS_PD = UNO_RS_PD;

// MF: phase 1 on v_mfma_f32_16x16x4_f32.  The dense 16 x NP row operator of the tile is the A operand (one LDS read per k-step
// from a table in operand layout), a lane's 16-byte piece of an input row is the B operand of FOUR column tiles (tile e = columns 4 n + e of the
// wave's 64 columns): one load instruction (4 rows x 256 contiguous bytes) feeds four MFMAs.  The VALU form spent 16 v_fmac and
// four LDS broadcast reads of the weights per loaded element - rocprofv3 on 446^2 -> 334^2: LDS 61 % busy, most of it those reads.
template <bool ACCUM, int KT, typename T, bool MF>      // KT: compiled tap count of the column operator (>= KW; weights past KW are zero)
// Occupancy (round 5): at ~84 registers the plain form ran 5 waves per SIMD = TWO 7-wave workgroups per CU, each alternating between a
// loading phase and a storing phase with 2 KB per wave in flight - 28 KB per CU, 3.0-3.3 TB/s on the down-sampling shapes.  Held to 64
// registers (no spills in the plain forms with up to 12 taps) four workgroups fit: tools/rsbench.py 446 -> 223 306 -> 281 us,
// 223 -> 111 160 -> 144, 334 -> 223 456 -> 410, the adjoints -5 .. -13 %.  The accumulating forms (16 old values per thread) spill at
// 64 registers and lose 5-12 %: with up to 5 taps (the up-sampling calls) they fit 72 registers - 7 waves - without spills (-1 .. -6 %),
// beyond that they keep the default, as does the plain form with 16 taps (11 spilled registers at 64).
__global__ __launch_bounds__(512, (ACCUM ? (KT <= 5 ? 7 : 1) : (KT <= 12 ? 8 : 1))) void resample_fused_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                             const int* __restrict__ tile_p0, const float* __restrict__ tile_w, int NP,
                                                             const int* __restrict__ startW, const float* __restrict__ wtW, int KW,
                                                             int H, int W, int Ho, int Wo, int n_img, int ntiles, int rev) {
    extern __shared__ __attribute__((aligned(16))) float V[];       // [RS_TR][WP] then (VALU form) the tile's dense weights [NP][16]
    // Workgroups go to the 8 XCDs round-robin by linear index: all row tiles of an image on ONE XCD, next to each other in time, so
    // that the band of input rows two neighbouring tiles share (NP - 16 H / Ho rows: 27 against 21.3 at 446 -> 334) comes out of
    // that XCD's L2 instead of over the fabric twice (FETCH_SIZE was 1.66x the image)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int n = (slot / ntiles) * 8 + xcd;
    const int tile = slot % ntiles;
    // rev: the images in DESCENDING order (alternating sweep direction, uno_common.h): the call runs right before the K1 of the same
    // block, which reads the same tensor - what this sweep read last is still in the Infinity Cache when K1, walking the other way, starts
    if (n >= n_img) return;
    if (rev) n = n_img - 1 - n;
    const int i0 = tile * RS_TR;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int p0 = tile_p0[tile];
    const int WP = (W + 3) & ~3;                                       // row pitch of V: 16-byte aligned rows
    float* sWd = V + RS_TR * WP;
    const T* src = in + (size_t)n * H * W;
    // phase-2 mapping (needed here already: the accumulating form requests the old values of its first column sweep before phase 1)
    T* dst = out + ((size_t)n * Ho + i0) * Wo;
    const int nr = min(RS_TR, Ho - i0);
    const int WoP = (Wo + 63) & ~63;
    const int G = max(1, nthreads / WoP);
    const int g = tid / WoP;
    const int jstride = G == 1 ? nthreads : WoP;
    const int j0 = tid - g * WoP;
    float old[ACCUM ? RS_TR : 1];
    // straight-line phase 2 (below) where the output rows are at least as long as the input rows (measured, tools/rsbench.py, old -> new:
    // accumulating 223 -> 446 490 -> 400 us, 111 -> 223 272 -> 206, 334 -> 446 595 -> 499, 223 -> 334 629 -> 508; plain 223 -> 446
    // 302 -> 277; the down-sampling shapes - half of the threads idle in phase 2 - LOSE 10-20 % with it and keep the rolled loop)
    const bool straight = G == 1 && Wo >= W;
    // down-sampling by two or more (the workgroup is at least twice as wide as an output row, rounded to half waves): TWO row groups of
    // 8 rows each, every thread one column, straight-line - no idle half of the workgroup, all LDS reads of a thread in flight together
    // (round 5; the rolled one-group loop ran these shapes at 2.8-3.7 TB/s: 446 -> 223, 223 -> 111 and their adjoints' columns)
    const int WoH = (Wo + 31) & ~31;
    const bool halves = !straight && nthreads >= 2 * WoH;
    const int gh = tid / WoH, jh = tid - gh * WoH;                       // halves: row group (0, 1; beyond: idle) and column
    if constexpr (ACCUM) {
        if (straight) {
            const int jc = min(j0, Wo - 1);
#pragma unroll
            for (int i = 0; i < RS_TR; ++i) old[i] = io_widen(dst[(size_t)min(i, nr - 1) * Wo + jc]);
        } else if (halves) {
            const int jc = min(jh, Wo - 1);
#pragma unroll
            for (int i = 0; i < RS_TR / 2; ++i) old[i] = io_widen(dst[(size_t)min(min(gh, 1) + 2 * i, nr - 1) * Wo + jc]);
        }
    }
    if constexpr (MF) {
        const int lane = tid & 63, n16 = lane & 15, kk = lane >> 4;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
        const int nks = (NP + 3) >> 2;
        // the row operator in A-operand layout: [k-step][lane (out row n16, k-slot kk)]
        // np_tile: rows of THIS tile's band (NP is the widest tile's; the table is zero beyond a tile's own band).  The rows a k-step
        // loads past it are replaced by zero below instead of meeting a zero weight: an Inf / NaN in a row just below a tile's band
        // must not become 0 x Inf = NaN in that tile's 16 output rows (the reference's banded interpolation never touches it)
        int& s_np = *reinterpret_cast<int*>(sWd + nks * 64);               // (one word of the dynamic allocation: a static __shared__ object on top of a 160 KB dynamic request fails)
        if (tid == 0) s_np = 0;
        __syncthreads();
        int my_np = 0;
        for (int e = tid; e < nks * 64; e += nthreads) {
            const int u = 4 * (e >> 6) + ((e & 63) >> 4);
            const float wv = u < NP ? tile_w[((size_t)tile * NP + u) * RS_TR + (e & 15)] : 0.f;
            sWd[e] = wv;
            if (wv != 0.f) my_np = max(my_np, u + 1);
        }
        if (my_np) atomicMax(&s_np, my_np);
        __syncthreads();
        const int np_tile = s_np;
        for (int c0 = 64 * wave; c0 < W; c0 += 64 * nwaves) {
            // this lane's four columns; pieces past the row end are pulled back inside the row (their results are not stored)
            const int col = min(c0 + 4 * n16, max(W - 4, 0));
            const T* colp = src + col;
            f32x4 acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = f32x4{0, 0, 0, 0};
            auto fetch = [&](int ks) { return io_ld4(colp + (size_t)min(p0 + 4 * ks + kk, H - 1) * W); };     // rows past the tile meet zero weights
            // RS_PD k-steps of rows in flight per lane
            float4 xb[RS_PD];
#pragma unroll
            for (int d = 0; d < RS_PD; ++d) xb[d] = fetch(min(d, nks - 1));
            for (int ks0 = 0; ks0 < nks; ks0 += RS_PD) {
#pragma unroll
                for (int d = 0; d < RS_PD; ++d) {
                    const int ks = ks0 + d;
                    if (ks < nks) {                                                      // uniform
                        const float a = sWd[ks * 64 + lane];
                        float4 x = xb[d];
                        // rows past the tile's band carry weight 0 - and are replaced by 0 here (see np_tile)
                        if (4 * ks + 3 >= np_tile) {                                     // uniform: the tile's last k-steps only
                            if (4 * ks + kk >= np_tile) x = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                        xb[d] = fetch(min(ks + RS_PD, nks - 1));                        // unconditional: the waits stay partial
                        acc[0] = mfma16(a, x.x, acc[0]);
                        acc[1] = mfma16(a, x.y, acc[1]);
                        acc[2] = mfma16(a, x.z, acc[2]);
                        acc[3] = mfma16(a, x.w, acc[3]);
                    }
                }
            }
            // accumulator register r of lane (g = kk, n16) of tile e: output row 4 g + r, column col + e
            if (c0 + 4 * n16 == col) {
                if (W >= 4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<f32x4*>(V + (4 * kk + r) * WP + col) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e < W) V[(4 * kk + r) * WP + col + e] = acc[e][r];
                }
            } else if (c0 + 4 * n16 < W) {
                // the row's last, shifted piece: columns c0 + 4 n16 .. W - 1 are its elements (c0 + 4 n16 - col) ..
                const int sh = c0 + 4 * n16 - col;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e >= sh) V[(4 * kk + r) * WP + col + e] = acc[e][r];
            }
        }
    } else {
    for (int e = tid; e < NP * RS_TR; e += nthreads) sWd[e] = tile_w[(size_t)tile * NP * RS_TR + e];   // [NP][16], zero outside the band
    __syncthreads();
    for (int q = tid; q < W; q += nthreads) {
        float acc[RS_TR];
#pragma unroll
        for (int r = 0; r < RS_TR; ++r) acc[r] = 0.f;
        // input rows in groups of 4, the next group's loads issued before the current group is consumed (the compiler's
        // own unrolling drained every group - s_waitcnt vmcnt(0) - before issuing the next: one memory latency per 4 rows)
        const T* col = src + q;
        auto load4 = [&](int u0, float x[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = io_widen(col[(size_t)min(p0 + u0 + i, H - 1) * W]);      // clamped: rows past the tile meet no weights
        };
        auto fma4 = [&](int u0, const float x[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (u0 + i < NP) {                                                             // uniform
                    const f32x4* w4 = reinterpret_cast<const f32x4*>(sWd + (u0 + i) * RS_TR);  // wave-uniform address: LDS broadcast
#pragma unroll
                    for (int r4 = 0; r4 < RS_TR / 4; ++r4) {
                        const f32x4 w = w4[r4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * r4 + e] = fmaf(w[e], x[i], acc[4 * r4 + e]);
                    }
                }
            }
        };
        float xa[4], xb[4];
        load4(0, xa);
        for (int u = 0; u < NP; u += 8) {
            load4(u + 4, xb);
            __builtin_amdgcn_sched_barrier(0);
            fma4(u, xa);
            load4(u + 8, xa);
            __builtin_amdgcn_sched_barrier(0);
            fma4(u + 4, xb);
        }
#pragma unroll
        for (int r = 0; r < RS_TR; ++r) V[r * WP + q] = acc[r];
    }
    }
    __syncthreads();
    // phase 2: thread -> (row group g, column j): when the workgroup is wider than a row, groups take rows round-robin.  The rows of a
    // thread are straight-line code (RS_TR / G of them at most; row index clamped for the reads, guarded for the stores): the LDS
    // reads of all rows are in flight together and the stores leave back to back.  ACCUM: the OLD values of a column's rows are
    // requested one column sweep ahead - the first sweep's before phase 1 (above) - so that no load round trip sits between
    // two stores (the rolled loop of rounds 1-3 paid one per row: load - wait - add - store, 16 times per thread and sweep)
    // one column sweep: NR = compile-time row count of a thread (16 when the workgroup is one row group, else the generic loop bound)
    auto sweep = [&](int j, auto g1) {
        constexpr bool G1 = decltype(g1)::value;
        const int s = startW[j];
        // straight-line taps: KW is a run-time value, and `if (t < KW)` per tap compiled to a chain of branches with a
        // load (then an LDS read) under each
        float w[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            float wv = wtW[(size_t)j * KW + min(t, KW - 1)];
            asm volatile("" : "+v"(wv));
            w[t] = t < KW ? wv : 0.f;
        }
        if constexpr (G1) {
            float cur[ACCUM ? RS_TR : 1];
            if constexpr (ACCUM) {
#pragma unroll
                for (int i = 0; i < RS_TR; ++i) cur[i] = old[i];
                const int jn = min(j + jstride, Wo - 1);           // next sweep's column (clamped: a valid address, unused past the end)
#pragma unroll
                for (int i = 0; i < RS_TR; ++i) old[i] = io_widen(dst[(size_t)min(i, nr - 1) * Wo + jn]);
            }
#pragma unroll
            for (int i = 0; i < RS_TR; ++i) {
                const float* v = V + i * WP;                        // (rows past nr hold finite values: phase 1 writes all 16)
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < KT; ++t) acc = fmaf(w[t], v[min(s + t, W - 1)], acc);
                // ACCUM is a template parameter: a run-time flag here put a conditional load into the store loop and
                // cost the plain path 60 % (227 -> 362 us at 1024 x 446^2 -> 223^2)
                if (i < nr) io_store1(dst + (size_t)i * Wo + j, ACCUM ? cur[i] + acc : acc);
            }
        } else {
            for (int r = g; r < nr; r += G) {
                const float* v = V + r * WP;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < KT; ++t) acc = fmaf(w[t], v[min(s + t, W - 1)], acc);
                io_store1(dst + (size_t)r * Wo + j, ACCUM ? io_widen(dst[(size_t)r * Wo + j]) + acc : acc);
            }
        }
    };
    if (halves) {
        if (gh < 2 && jh < Wo) {
            const int j = jh, s = startW[j];
            float w[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                float wv = wtW[(size_t)j * KW + min(t, KW - 1)];
                asm volatile("" : "+v"(wv));
                w[t] = t < KW ? wv : 0.f;
            }
#pragma unroll
            for (int ii = 0; ii < RS_TR / 2; ++ii) {
                const int i = gh + 2 * ii;
                const float* v = V + i * WP;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < KT; ++t) acc = fmaf(w[t], v[min(s + t, W - 1)], acc);
                if (i < nr) io_store1(dst + (size_t)i * Wo + j, ACCUM ? old[ii] + acc : acc);
            }
        }
    } else if (straight) {
        for (int j = j0; j < Wo; j += jstride) sweep(j, std::true_type{});
    } else if (g < G) {
        for (int j = j0; j < Wo; j += jstride) sweep(j, std::false_type{});
    }
}

// one instantiation of the fused kernel: raises its dynamic-LDS limit when a long row needs more than the 64 KB a launch gets by
// default (rows of 1000 .. 2400 floats: the 1024^2 / 1089^2 levels of config C5), then launches
// (round 4, measured and not kept: PERSISTENT workgroups walking the (image, tile) slots of their XCD - as many workgroups as the
// device holds at once - are 5-45 % slower on every shape of tools/rsbench.py than one workgroup per tile)
template <bool A, int K, typename T, bool MF>
static bool launch_fused_one(dim3 grid, int nthreads, size_t lds, hipStream_t s, const T* in, T* out, const int* tile_p0, const float* tile_w,
                             int NP, const int* startW, const float* wtW, int KW, int H, int W, int Ho, int Wo, int n_img, int ntiles, int rev) {
    auto k = resample_fused_kernel<A, K, T, MF>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_slot)) return false;
    hipLaunchKernelGGL(k, grid, dim3(nthreads), lds, s, in, out, tile_p0, tile_w, NP, startW, wtW, KW, H, W, Ho, Wo, n_img, ntiles, rev);
    return true;
}

int launch_resample2d(const void* in_, void* out_, float* tmp, int n_img, int H, int W, int Ho, int Wo, const int* startH,
                      const float* wtH, int KH, const int* startW, const float* wtW, int KW, const int* tile_p0,
                      const float* tile_w, int NP, int accumulate, int bf16, hipStream_t s) {
    accumulate &= 1;                                // (bit 1, the explicit "descending" request of this round's first form, is superseded)
    const int rev = next_sweep_reversed(SWEEP_K7);          // fused kernel: images in descending order on every other launch (uno_common.h)
    typedef unsigned short bf_t;
    const float* in = static_cast<const float*>(in_);
    float* out = static_cast<float*>(out_);
    const bf_t* inb = static_cast<const bf_t*>(in_);
    bf_t* outb = static_cast<bf_t*>(out_);
    const double es = bf16 ? 2.0 : 4.0;
    if (KH < 1 || KW < 1) { set_error("resample2d: empty band (%d, %d)", KH, KW); return -2; }
    const bool wide_band = KW > 16;         // the column operator's taps live in registers up to 16; beyond that: the generic two-pass form
    // dynamic LDS of the fused kernel: the 16 x W tile (rows padded to 4 floats) + the dense row-operator tile.  Gated on the
    // REAL request (W ~ 980..1024 passes a tile-only test and then asks for more than the 64 KB a launch gets without the
    // dynamic-LDS attribute: the launch fails instead of falling through to the two-pass form)
    const size_t lds = (size_t)RS_TR * ((W + 3) & ~3) * sizeof(float) + (size_t)((NP + 3) / 4) * 64 * sizeof(float) + 16;
    // (round 3: up to 150 KB - one workgroup per CU - through the dynamic-LDS attribute; the two-pass form took 2.0 ms per call at
    // the 1089 -> 544 level of the C5 model where this kernel needs 0.5)
    if (!wide_band && tile_p0 && tile_w && NP >= 1 && NP <= 96 && lds <= 150 * 1024) {
        // fused single-pass kernel: needs the dense row-tile tables and the 16 x W tile to fit in LDS
        ProfScope prof("uno::resample_fused_kernel", es * n_img * ((double)H * W + (accumulate ? 2.0 : 1.0) * Ho * Wo), s);
        // one column per thread in phase 1: the narrowest multiple of 64 threads that covers W in whole sweeps
        const int sweeps = (W + 511) / 512;
        const int nthreads = ((((W + sweeps - 1) / sweeps) + 63) / 64) * 64;
        const int ntiles = (Ho + RS_TR - 1) / RS_TR;
        const dim3 grid((unsigned)(((n_img + 7) / 8) * 8) * ntiles);
        const bool mf = W >= 4;             // row operator on MFMA (rows of at least one 16-byte piece)
        bool lds_ok = true;
#define UNO_RS_LAUNCH(A, K)                                                                                                    \
        do {                                                                                                                   \
            if (bf16 && mf) lds_ok = launch_fused_one<A, K, bf_t, true>(grid, nthreads, lds, s, inb, outb, tile_p0, tile_w, NP, startW, wtW, KW, H, W, Ho, Wo, n_img, ntiles, rev); \
            else if (bf16) lds_ok = launch_fused_one<A, K, bf_t, false>(grid, nthreads, lds, s, inb, outb, tile_p0, tile_w, NP, startW, wtW, KW, H, W, Ho, Wo, n_img, ntiles, rev); \
            else if (mf) lds_ok = launch_fused_one<A, K, float, true>(grid, nthreads, lds, s, in, out, tile_p0, tile_w, NP, startW, wtW, KW, H, W, Ho, Wo, n_img, ntiles, rev); \
            else lds_ok = launch_fused_one<A, K, float, false>(grid, nthreads, lds, s, in, out, tile_p0, tile_w, NP, startW, wtW, KW, H, W, Ho, Wo, n_img, ntiles, rev); \
        } while (0)
        // tap counts seen in the U-NO models: 4-5 (up-sampling by ~2 and its adjoint's rows), 9-10 (down-sampling by ~2)
#define UNO_RS_PICK(A)                                                                                                         \
        do {                                                                                                                   \
            if (KW <= 4) UNO_RS_LAUNCH(A, 4); else if (KW <= 5) UNO_RS_LAUNCH(A, 5); else if (KW <= 8) UNO_RS_LAUNCH(A, 8);    \
            else if (KW <= 9) UNO_RS_LAUNCH(A, 9); else if (KW <= 10) UNO_RS_LAUNCH(A, 10);                                    \
            else if (KW <= 12) UNO_RS_LAUNCH(A, 12); else UNO_RS_LAUNCH(A, 16);                                                \
        } while (0)
        if (accumulate) UNO_RS_PICK(true); else UNO_RS_PICK(false);
#undef UNO_RS_PICK
#undef UNO_RS_LAUNCH
        if (!lds_ok) { set_error("resample2d: cannot raise dynamic LDS to %zu", lds); return -4; }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("resample2d launch: %s", hipGetErrorString(e)); return -5; }
        return 0;
    }
    const int RPB = 16;
    // rows first when that shrinks the intermediate (Ho*W <= H*Wo), columns first otherwise
    const bool rows_first = (long long)Ho * W <= (long long)H * Wo;
    const double img_bytes = 4.0 * n_img;       // the intermediate is f32 in both element types
    if (rows_first) {
        const dim3 g1((W + 255) / 256, (Ho + RPB - 1) / RPB, n_img), g2((Wo + 255) / 256, (Ho + RPB - 1) / RPB, n_img);
        {
            ProfScope prof("uno::resample_rows_kernel", n_img * (es * H * W + 4.0 * Ho * W), s);
            if (bf16) hipLaunchKernelGGL((resample_rows_kernel<bf_t, float>), g1, dim3(256), 0, s, inb, tmp, startH, wtH, KH, H, Ho, W, RPB, 0);
            else hipLaunchKernelGGL((resample_rows_kernel<float, float>), g1, dim3(256), 0, s, in, tmp, startH, wtH, KH, H, Ho, W, RPB, 0);
        }
        {
            ProfScope prof("uno::resample_cols_kernel", n_img * (4.0 * Ho * W + es * Ho * Wo), s);
            if (wide_band && bf16) hipLaunchKernelGGL((resample_cols_generic_kernel<float, bf_t>), g2, dim3(256), 0, s, (const float*)tmp, outb, startW, wtW, KW, Ho, W, Wo, RPB, accumulate);
            else if (wide_band) hipLaunchKernelGGL((resample_cols_generic_kernel<float, float>), g2, dim3(256), 0, s, (const float*)tmp, out, startW, wtW, KW, Ho, W, Wo, RPB, accumulate);
            else if (bf16) hipLaunchKernelGGL((resample_cols_kernel<float, bf_t>), g2, dim3(256), 0, s, (const float*)tmp, outb, startW, wtW, KW, Ho, W, Wo, RPB, accumulate);
            else hipLaunchKernelGGL((resample_cols_kernel<float, float>), g2, dim3(256), 0, s, (const float*)tmp, out, startW, wtW, KW, Ho, W, Wo, RPB, accumulate);
        }
    } else {
        const dim3 g1((Wo + 255) / 256, (H + RPB - 1) / RPB, n_img), g2((Wo + 255) / 256, (Ho + RPB - 1) / RPB, n_img);
        {
            ProfScope prof("uno::resample_cols_kernel", n_img * (es * H * W + 4.0 * H * Wo), s);
            if (wide_band && bf16) hipLaunchKernelGGL((resample_cols_generic_kernel<bf_t, float>), g1, dim3(256), 0, s, inb, tmp, startW, wtW, KW, H, W, Wo, RPB, 0);
            else if (wide_band) hipLaunchKernelGGL((resample_cols_generic_kernel<float, float>), g1, dim3(256), 0, s, in, tmp, startW, wtW, KW, H, W, Wo, RPB, 0);
            else if (bf16) hipLaunchKernelGGL((resample_cols_kernel<bf_t, float>), g1, dim3(256), 0, s, inb, tmp, startW, wtW, KW, H, W, Wo, RPB, 0);
            else hipLaunchKernelGGL((resample_cols_kernel<float, float>), g1, dim3(256), 0, s, in, tmp, startW, wtW, KW, H, W, Wo, RPB, 0);
        }
        {
            ProfScope prof("uno::resample_rows_kernel", n_img * (4.0 * H * Wo + es * Ho * Wo), s);
            if (bf16) hipLaunchKernelGGL((resample_rows_kernel<float, bf_t>), g2, dim3(256), 0, s, (const float*)tmp, outb, startH, wtH, KH, H, Ho, Wo, RPB, accumulate);
            else hipLaunchKernelGGL((resample_rows_kernel<float, float>), g2, dim3(256), 0, s, (const float*)tmp, out, startH, wtH, KH, H, Ho, Wo, RPB, accumulate);
        }
    }
    (void)img_bytes;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("resample2d launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
