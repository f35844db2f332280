// K2/K4 - batched per-mode complex GEMM on the truncated spectrum.
//
//   out(m, n, p) = sum_k A'(m, k, p) * B'(k, n, p)          (' = optional complex conjugate)
//
// One kernel serves the three contractions of SpectralConv{2,3}d_Uno (reference
// integral_operators.py:178-179 / :382-383 and their autograd adjoints):
//   forward   O[b,o]  = sum_i X[b,i]        * W[i,o]          einsum "bixy,ioxy->boxy"
//   grad X    gX[b,i] = sum_o gO[b,o]       * conj(W[i,o])
//   grad W    gW[i,o] = sum_b conj(X[b,i])  * gO[b,o]
// by choosing operand strides (ModeGemmParams).  Modes p are independent; they are split into
// `ncorner` contiguous runs of Mc modes (one run per weight tensor).
//
// Workgroup tile: 16 (m) x 16 (n) x QC = 16 consecutive modes, K streamed in chunks of KC = 8.
// Both operands are read from HBM/L2 in 128-byte runs along the mode axis (the minor axis of the
// reference's (Ci, Co, m1, m2) parameter layout), transposed through LDS into per-mode
// [k][16] planes (re / im), and consumed as v_mfma_f32_16x16x4_f32 fragments with conflict-free
// ds_read_b32.  A complex product is four real MFMAs.  The 16x16x16-mode result tile goes back
// through LDS so the stores are 128-byte runs again.  The next K chunk is prefetched into registers
// while the current one is multiplied.
#include "uno_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace uno {

constexpr int KC = 8;           // reduction chunk staged in LDS

// QC = modes per workgroup: 16 (128-byte runs along the mode axis) or 8 (64-byte runs, twice the workgroups: layers with few
// modes and many channels - 2 x 64 modes x 256 x 256 channels - give only 128 workgroups of 16 modes, and a workgroup's phases
// (stage to LDS, issue loads, multiply) do not overlap with one wave per SIMD: the time was the sum of the three)
// PIPE: software-pipelined K loop over two LDS buffers (8-mode variant on layers with few mode chunks, see the launcher)
// BH: operand B (the weights of ops 0 / 1) is stored as half-precision (re, im) pairs - config C5's weight storage - and widened
// as it is loaded; everything after the load is the complex64 kernel.
// ACC: out += (weight gradients added into a parameter's gradient buffer).  A template parameter, so that the plain kernels are
// the same code as before the accumulating form existed (a run-time flag in the store loops cost the plain weight gradient of
// the C2 / C4 blocks 20-40 %: 30 -> 37 us, and with both loops in one kernel the pointer arrays went to scratch)
template <int QC, bool PIPE, bool BH, bool ACC = false>
__global__ __launch_bounds__(256) void mode_gemm_kernel(ModeGemmParams p) {
    constexpr int TILE_ELEMS = 16 * KC * QC;        // complex elements of one operand chunk
    constexpr int EPT = TILE_ELEMS / 256;           // elements per thread per operand: 8 / 4
    constexpr int KSTEP = 16 / QC;                  // k rows covered by one pass of the 256 threads: 1 / 2
    constexpr int PLANE = KC * 16 + 64 / QC;        // floats per (mode) plane; the pad makes the transposing writes conflict-free (lane -> bank 4 q + x / 8 q + x)
    // two buffers of [operand A|B][re|im][QC][PLANE] floats (chunk c is multiplied out of one while chunk c + 1 is staged into
    // the other: one barrier per chunk); reused as the [16 m][16 n][QC] c64 output tile
    constexpr int SB = 2 * 2 * QC * PLANE;
    static_assert((PIPE ? 2 : 1) * SB >= 16 * 16 * (QC + 1) * 2 || !PIPE, "output tile must fit the staging buffers");
    extern __shared__ __attribute__((aligned(16))) float sm[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;

    const int nq = (p.Mc + QC - 1) / QC;
    const int corner = blockIdx.x / nq;
    const int q0 = (blockIdx.x % nq) * QC;
    const int n0 = blockIdx.y * 16;
    const int m0 = blockIdx.z * 16;
    const int nmodes = min(QC, p.Mc - q0);

    const float2* Ab = p.A.base[corner] + q0;
    constexpr int ESB = BH ? 4 : 8;         // bytes per complex element of operand B
    const char* Bb = reinterpret_cast<const char*>(p.B.base[corner]) + (size_t)q0 * ESB;
    const float sgnA = p.A.conj ? -1.f : 1.f;
    const float sgnB = p.B.conj ? -1.f : 1.f;

    // staging map: element e = tid + 256 * u  ->  (row = e / QC, q = e % QC); A rows = (k, m), B rows = (k, n).  With 256 threads
    // this is q = tid % QC, x = (tid / QC) % 16, k-local = KSTEP * u + tid / (16 QC): a thread walks K with a fixed (x, q), so
    // its addresses are one base per operand plus k * stride (computing the general form per load cost 20-40 integer
    // instructions each)
    float2 ra[2][EPT], rb[2][EPT];           // two register stages: loads run two chunks ahead of the multiply
    const int q_t = tid & (QC - 1), x_t = (tid / QC) & 15, kb_t = tid / (16 * QC);
    const bool okA = q_t < nmodes && m0 + x_t < p.M, okB = q_t < nmodes && n0 + x_t < p.N;
    // raw buffer loads: per-thread byte offset (fixed) + scalar offset k * stride - no per-load vector address arithmetic
    // (the 64-bit address of each of the 16 loads of a chunk cost 3-4 VALU instructions, issued between the MFMA blocks)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, -1, 0x00020000);
    const unsigned voA = okA ? (unsigned)(((long long)(m0 + x_t) * p.A.s0 + q_t + (long long)kb_t * p.A.s1) * 8) : 0u;
    const unsigned voB = okB ? (unsigned)(((long long)(n0 + x_t) * p.B.s1 + q_t + (long long)kb_t * p.B.s0) * ESB) : 0u;
    const unsigned strideA = (unsigned)(p.A.s1 * 8), strideB = (unsigned)(p.B.s0 * ESB);
    auto load_chunk = [&](float2* da, float2* db, int k0) {
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            // clamped (wave-uniform) row: unconditional loads; the zero-fill of invalid entries happens on the way to LDS,
            // after the MFMA block (masking here would consume the registers at once)
            const unsigned kb = (unsigned)max(min(k0 + KSTEP * u, p.K - KSTEP), 0);
            const u32x2 ta = __builtin_amdgcn_raw_buffer_load_b64(rA, voA, kb * strideA, 0);
            da[u] = make_float2(__uint_as_float(ta[0]), __uint_as_float(ta[1]));
            if constexpr (BH) {
                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                const unsigned tb = __builtin_amdgcn_raw_buffer_load_b32(rB, voB, kb * strideB, 0);
                const h2_t hv = __builtin_bit_cast(h2_t, tb);
                db[u] = make_float2((float)hv[0], (float)hv[1]);
            } else {
                const u32x2 tb = __builtin_amdgcn_raw_buffer_load_b64(rB, voB, kb * strideB, 0);
                db[u] = make_float2(__uint_as_float(tb[0]), __uint_as_float(tb[1]));
            }
        }
    };
    auto store_chunk = [&](float* buf, const float2* sa, const float2* sb, int k0) {
        float* sA = buf;
        float* sB = buf + 2 * QC * PLANE;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int o = q_t * PLANE + (KSTEP * u + kb_t) * 16 + x_t;     // row = k-local * 16 + x
            const bool kv = k0 + KSTEP * u + kb_t < p.K;
            const float2 va = (kv && okA) ? sa[u] : make_float2(0.f, 0.f);
            const float2 vb = (kv && okB) ? sb[u] : make_float2(0.f, 0.f);
            sA[o] = va.x; sA[QC * PLANE + o] = sgnA * va.y;
            sB[o] = vb.x; sB[QC * PLANE + o] = sgnB * vb.y;
        }
    };

    f32x4 accr[QC / 4], acci[QC / 4];       // this wave's modes q = wave + 4 * v
#pragma unroll
    for (int v = 0; v < QC / 4; ++v) { accr[v] = f32x4{0, 0, 0, 0}; acci[v] = f32x4{0, 0, 0, 0}; }

    auto multiply = [&](const float* buf) {
        const float* sA = buf;
        const float* sB = buf + 2 * QC * PLANE;
#pragma unroll
        for (int v = 0; v < QC / 4; ++v) {
            const int q = wave + 4 * v;
            const float* pa = sA + q * PLANE + kk * 16 + r16;
            const float* pb = sB + q * PLANE + kk * 16 + r16;
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                const float ar = pa[ks * 64], ai = pa[QC * PLANE + ks * 64];
                const float br = pb[ks * 64], bi = pb[QC * PLANE + ks * 64];
                accr[v] = mfma16(ar, br, accr[v]);
                acci[v] = mfma16(ar, bi, acci[v]);
                accr[v] = mfma16(-ai, bi, accr[v]);
                acci[v] = mfma16(ai, br, acci[v]);
            }
        }
    };
    // Software pipeline over K chunks (straight-line stages, all loads unconditional with clamped rows, so the waits are
    // s_waitcnt vmcnt(16) - not 0): while chunk c is multiplied out of LDS buffer c & 1, chunk c + 1 (in registers since the
    // previous stage) is staged into the other buffer and the loads of chunk c + 2 are in flight.  Chunks past K hold zeros.
    float* buf0 = sm;
    float* buf1 = sm + SB;
    if constexpr (PIPE) {
        load_chunk(ra[0], rb[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(ra[1], rb[1], KC);
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(buf0, ra[0], rb[0], 0);
        __syncthreads();
        for (int k0 = 0; k0 < p.K; k0 += 2 * KC) {
            load_chunk(ra[0], rb[0], k0 + 2 * KC);
            __builtin_amdgcn_sched_barrier(0);
            multiply(buf0);
            store_chunk(buf1, ra[1], rb[1], k0 + KC);
            __syncthreads();
            load_chunk(ra[1], rb[1], k0 + 3 * KC);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + KC < p.K) multiply(buf1);
            store_chunk(buf0, ra[0], rb[0], k0 + 2 * KC);
            __syncthreads();
        }
    } else {
        // one buffer, the next chunk in registers while the current one is multiplied.  For 16 modes the pipelined form needs
        // 68 KB of LDS (2 workgroups per CU instead of 4) and measured 15-25 % slower on the large grids that variant serves.
        load_chunk(ra[0], rb[0], 0);
        for (int k0 = 0; k0 < p.K; k0 += KC) {
            __syncthreads();                    // previous chunk fully consumed
            store_chunk(buf0, ra[0], rb[0], k0);
            __syncthreads();
            if (k0 + KC < p.K) load_chunk(ra[0], rb[0], k0 + KC);
            multiply(buf0);
        }
        __syncthreads();
    }

    // result tile -> LDS as [m][n][QC+1] c64, then 128-byte runs along the mode axis
    float2* sO = reinterpret_cast<float2*>(sm);
#pragma unroll
    for (int v = 0; v < QC / 4; ++v) {
        const int q = wave + 4 * v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * kk + r;
            sO[(m * 16 + r16) * (QC + 1) + q] = make_float2(accr[v][r], acci[v][r]);
        }
    }
    __syncthreads();
    float2* Ob = p.out[corner] + q0;
    if constexpr (!ACC) {
        for (int e = tid; e < 16 * 16 * QC; e += 256) {
            const int q = e & (QC - 1), mn = e / QC;
            const int m = mn >> 4, n = mn & 15;
            if (q < nmodes && m0 + m < p.M && n0 + n < p.N)
                Ob[(long long)(m0 + m) * p.o_sm + (long long)(n0 + n) * p.o_sn + q] = sO[mn * (QC + 1) + q];
        }
    } else {
        // QC iterations per thread (16 x 16 x QC / 256) in groups of four: the old values of a group are requested before its
        // first store (element by element every store waits for its own read: the compiler may not move a load above an earlier
        // store to the same array)
        for (int e0 = tid; e0 < 16 * 16 * QC; e0 += 4 * 256) {
            float2* dst[4];
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u;
                const int q = e & (QC - 1), mn = e / QC;
                const int m = mn >> 4, n = mn & 15;
                const bool ok = e < 16 * 16 * QC && q < nmodes && m0 + m < p.M && n0 + n < p.N;
                dst[u] = ok ? Ob + (long long)(m0 + m) * p.o_sm + (long long)(n0 + n) * p.o_sn + q : nullptr;
                v[u] = ok ? sO[mn * (QC + 1) + q] : make_float2(0.f, 0.f);
                if (ok) { const float2 o = *dst[u]; v[u].x += o.x; v[u].y += o.y; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u]) *dst[u] = v[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- K2b
// The same contraction on v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 outer products per instruction, one per MODE.
// Lane 4 q + x supplies A[m = 4 mt + x][k] and B[k][n = 4 nt + x] of mode q0 + q, and the mode axis is the contiguous axis of all
// three operands - so an 8-byte load per lane (4 rows x 128 contiguous bytes per instruction) IS the MFMA operand and the
// accumulator register i of lane 4 q + j IS out[m = 4 mt + i][n = 4 nt + j][q0 + q]: no LDS staging, no transposes, no barriers
// in the K loop, and modes / rows past the end never mix with valid ones (their blocks / rows are simply not stored).
// A wave owns MTW x NTW tiles of 4 x 4 outputs for 16 modes and walks K with its loads PF steps ahead; layers with few
// (mode, tile) tasks split K over the waves of a workgroup and reduce through LDS.
__device__ __forceinline__ f32x4 mfma4b(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);      // A[i] = lane 4 blk + i, B[j] = lane 4 blk + j, D[i][j] = lane 4 blk + j, reg i
}


// The MFMA wants the four rows of a block in four CONSECUTIVE lanes, memory wants consecutive lanes on consecutive addresses (with
// lane 4 q + x on row x the address unit sees four different 128-byte lines in every group of four lanes: measured 27 us at the
// Darcy block against 17 us for the LDS-staged kernel).  So lane 16 x + q loads / stores row x, mode q - 16 consecutive lanes on
// 128 contiguous bytes - and a 64-lane transpose (ds_bpermute_b32: the LDS crossbar, no LDS memory) moves the value to lane 4 q + x.
__device__ __forceinline__ float lane_pull(int src_lane_x4, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane_x4, __float_as_int(v)));
}

// adjacent_modes: the (4 / KS) wave groups of a workgroup take ADJACENT MODE GROUPS of the same columns instead of adjacent column groups of
// one mode group - their 128-byte pieces are then neighbours in memory (512-byte runs requested together).  Round 5 sweep (tools/dev/k2dev.py:
// twenty layer shapes x three roles x KS x this switch, cold operands, every configuration checked against the float64 einsum): it pays
// on the 3-D layers with ~1000 modes or more per corner and enough channels to fill the chip with such workgroups - C4 block forward
// 27.2 -> 20.8 us, input gradient 28.2 -> 22.8, weight gradient 23.3 -> 19.9; Uno3D_T20 (width 32) layer 1 65 -> 57 / 65 -> 59 / 71 -> 62,
// layer 2 81 -> 75 / 83 -> 69 - and loses everywhere else (C2 block 17 -> 26, 256 x 256 channels at 64 modes weight gradient 25 -> 76).
template <int MTW, int NTW, int K2B_PF, bool BH, bool ACC = false>
__device__ __forceinline__ void mode_gemm_blocks_body(const ModeGemmParams& p, int KS, int ngw, int per_group, int adjacent_modes, int bid, float* smb) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane >> 2, x = lane & 3;                          // MFMA layout: block (mode) q, row / column x
    const int mq = lane & 15, mx = lane >> 4;                       // memory layout: mode mq, row / column mx
    const int to_mfma = 4 * (16 * x + q), to_mem = 4 * (4 * mq + mx);      // ds_bpermute source lanes (x 4 bytes)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups go to the 8 XCDs round-robin by linear index, and every workgroup of a mode group reads the same A rows (all
    // column groups) or the same B rows (all row groups): keep a mode group on ONE XCD, so that its operands come out of that
    // XCD's L2 instead of being fetched over the fabric once per workgroup.
    const int nq = (p.Mc + 15) >> 4;
    const int xcd = bid & 7, slot = bid >> 3;
    int grp = (slot / per_group) * 8 + xcd;                                        // mode group (adjacent_modes: group of 4 / KS)
    const int within = slot % per_group;                                           // (column-group workgroup, row group) inside it
    if (adjacent_modes) grp = (4 / KS) * grp + wave / KS;
    if (grp >= p.ncorner * nq) return;
    const int corner = grp / nq, q0 = (grp % nq) * 16;
    const int kidx = wave % KS, ng = adjacent_modes ? within % ngw : (within % ngw) * (4 / KS) + wave / KS;
    const int n0 = ng * 4 * NTW, m0 = (within / ngw) * 4 * MTW;
    const bool active = n0 < p.N;
    constexpr int ESB = BH ? 4 : 8;
    const int pq = min(q0 + mq, p.Mc - 1);                         // modes past the end: a clamped (valid) address, never stored
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A.base[corner], 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B.base[corner], 0, -1, 0x00020000);
    unsigned voA[MTW], voB[NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) voA[mt] = (unsigned)(((long long)min(m0 + 4 * mt + mx, p.M - 1) * p.A.s0 + pq) * 8);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) voB[nt] = (unsigned)(((long long)min(n0 + 4 * nt + mx, p.N - 1) * p.B.s1 + pq) * ESB);
    const unsigned strideA = (unsigned)(p.A.s1 * 8), strideB = (unsigned)(p.B.s0 * ESB);
    const float sgnA = p.A.conj ? -1.f : 1.f, sgnB = p.B.conj ? -1.f : 1.f;
    const int kper = (p.K + KS - 1) / KS, kb = kidx * kper, ke = min(p.K, kb + kper);

    f32x4 Dr[MTW][NTW], Di[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) { Dr[mt][nt] = f32x4{0, 0, 0, 0}; Di[mt][nt] = f32x4{0, 0, 0, 0}; }

    float2 ra[K2B_PF][MTW], rb[K2B_PF][NTW];          // as loaded (memory layout)
    float2 pa[2][MTW], pb[2][NTW];                    // transposed to the MFMA layout, one step ahead of the multiply
    auto load_step = [&](int s, int k) {
        const unsigned kk = (unsigned)min(k, p.K - 1);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rA, voA[mt], kk * strideA, 0);
            ra[s][mt] = make_float2(__uint_as_float(t[0]), __uint_as_float(t[1]));
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            if constexpr (BH) {
                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(rB, voB[nt], kk * strideB, 0);
                const h2_t hv = __builtin_bit_cast(h2_t, t);
                rb[s][nt] = make_float2((float)hv[0], (float)hv[1]);
            } else {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rB, voB[nt], kk * strideB, 0);
                rb[s][nt] = make_float2(__uint_as_float(t[0]), __uint_as_float(t[1]));
            }
        }
    };
    auto transpose_step = [&](int s, int d) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) pa[d][mt] = make_float2(lane_pull(to_mfma, ra[s][mt].x), lane_pull(to_mfma, ra[s][mt].y));
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) pb[d][nt] = make_float2(lane_pull(to_mfma, rb[s][nt].x), lane_pull(to_mfma, rb[s][nt].y));
    };
    // Register stage s holds step k + s.  Per step: multiply the transposed operands of this step, refill the stage they came from
    // (unconditionally, clamped row), transpose the next step's stage - whose loads were issued PF - 1 steps ago, so the waits in
    // the loop are s_waitcnt vmcnt((PF - 1) steps' loads), never 0.  Steps past the end of this wave's K range replace the (valid,
    // clamped) row by zero: one straight-line loop body, no tail code.
    if (active && kb < ke) {
#pragma unroll
        for (int s = 0; s < K2B_PF; ++s) {
            load_step(s, kb + s);
            __builtin_amdgcn_sched_barrier(0);           // issue order = stage order, in the prologue as in the loop
        }
        transpose_step(0, 0);
#pragma unroll 1
        for (int k = kb; k < ke; k += K2B_PF) {
#pragma unroll
            for (int s = 0; s < K2B_PF; ++s) {
                const int d = s & 1;
                const bool live = k + s < ke;          // (a select, not a multiplication by 0 / 1: the clamped row may hold Inf / NaN)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const float ar = live ? pa[d][mt].x : 0.f, ai = live ? sgnA * pa[d][mt].y : 0.f, nai = -ai;
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const float br = pb[d][nt].x, bi = sgnB * pb[d][nt].y;
                        Dr[mt][nt] = mfma4b(ar, br, Dr[mt][nt]);
                        Di[mt][nt] = mfma4b(ar, bi, Di[mt][nt]);
                        Dr[mt][nt] = mfma4b(nai, bi, Dr[mt][nt]);
                        Di[mt][nt] = mfma4b(ai, br, Di[mt][nt]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                load_step(s, k + s + K2B_PF);
                __builtin_amdgcn_sched_barrier(0);
                transpose_step((s + 1) % K2B_PF, d ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    float2* Ob = p.out[corner];
    const bool qv = q0 + mq < p.Mc;
    // accumulator register i of MFMA lane 4 q + j is out[m = 4 mt + i][n = 4 nt + j][q0 + q]; lane 16 j + q stores it
    auto store_tile = [&](int mt, int nt, const f32x4& vr, const f32x4& vi) {
        const int n = n0 + 4 * nt + mx;
        if constexpr (!ACC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 v = make_float2(lane_pull(to_mem, vr[i]), lane_pull(to_mem, vi[i]));
                const int m = m0 + 4 * mt + i;
                if (qv && n < p.N && m < p.M) Ob[(long long)m * p.o_sm + (long long)n * p.o_sn + q0 + mq] = v;
            }
        } else {
            // the tile's four old values first, then its four stores.  The lane exchange (ds_bpermute) runs with ALL lanes enabled,
            // outside the guards: a lane switched off by a guard supplies nothing to the lane that pulls from it
            float2 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = make_float2(lane_pull(to_mem, vr[i]), lane_pull(to_mem, vi[i]));
            float2* dst[4];
            float2 old[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 4 * mt + i;
                dst[i] = (qv && n < p.N && m < p.M) ? Ob + (long long)m * p.o_sm + (long long)n * p.o_sn + q0 + mq : nullptr;
                old[i] = dst[i] ? *dst[i] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (dst[i]) *dst[i] = make_float2(v[i].x + old[i].x, v[i].y + old[i].y);
        }
    };
    if (KS == 1) {
        if (!active) return;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) store_tile(mt, nt, Dr[mt][nt], Di[mt][nt]);
        return;
    }
    // K split over KS waves: every wave leaves its partial tiles in LDS as [wave][tile][re | im][reg][lane]; wave r of a split group
    // then sums and stores the tiles t = r (mod KS)
    constexpr int TILES = MTW * NTW;
    float* mine = smb + (size_t)wave * TILES * 8 * 64;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mine[((mt * NTW + nt) * 8 + i) * 64 + lane] = Dr[mt][nt][i];
                mine[((mt * NTW + nt) * 8 + 4 + i) * 64 + lane] = Di[mt][nt][i];
            }
    __syncthreads();
    if (!active) return;
    const float* grpb = smb + (size_t)(wave - kidx) * TILES * 8 * 64;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        if (t % KS != kidx) continue;
        f32x4 vr = f32x4{0, 0, 0, 0}, vi = f32x4{0, 0, 0, 0};
        for (int w = 0; w < KS; ++w)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vr[i] += grpb[((size_t)w * TILES * 8 + t * 8 + i) * 64 + lane];
                vi[i] += grpb[((size_t)w * TILES * 8 + t * 8 + 4 + i) * 64 + lane];
            }
        store_tile(t / NTW, t % NTW, vr, vi);
    }
}

template <int MTW, int NTW, int K2B_PF, bool BH, bool ACC = false>
__global__ __launch_bounds__(256) void mode_gemm_blocks_kernel(ModeGemmParams p, int KS, int ngw, int per_group, int adjacent_modes) {
    extern __shared__ __attribute__((aligned(16))) float smb[];
    mode_gemm_blocks_body<MTW, NTW, K2B_PF, BH, ACC>(p, KS, ngw, per_group, adjacent_modes, (int)blockIdx.x, smb);
}

// The two GEMMs of a backward pass in ONE launch (reference: the autograd adjoints of integral_operators.py:178-179): workgroups
// [0, Ga) run the input-gradient role gX[b,i] = sum_o gO[b,o] conj(W[i,o]), the rest the weight-gradient role gW[i,o] = sum_b conj(X[b,i])
// gO[b,o].  Each role alone fills a fraction of the chip (1.3 - 2.4 TB/s in the training step, profiles/r05_step_launches.txt) and its
// launch ends in a tail of a few workgroups; together they share the read-only gO out of the same XCD's L2 (a mode group lands on the
// same XCD in both roles) and one launch's ramp and tail.  The bodies are the single-role kernel's, unchanged.
struct BlocksCfg { int KS, ngw, per_group, adjacent; };
template <int MA, int NA, int PFA, bool ACCB>
__global__ __launch_bounds__(256) void mode_gemm_blocks_pair_kernel(ModeGemmParams pa, BlocksCfg ca, int Ga, ModeGemmParams pb, BlocksCfg cb) {
    extern __shared__ __attribute__((aligned(16))) float smb[];
    if ((int)blockIdx.x < Ga) mode_gemm_blocks_body<MA, NA, PFA, false, false>(pa, ca.KS, ca.ngw, ca.per_group, ca.adjacent, (int)blockIdx.x, smb);
    else mode_gemm_blocks_body<4, 4, 2, false, ACCB>(pb, cb.KS, cb.ngw, cb.per_group, cb.adjacent, (int)blockIdx.x - Ga, smb);
}

template <int MTW, int NTW, int PF>
static void launch_blocks_t(const ModeGemmParams& p, int KS, int adjacent_modes, hipStream_t s) {
#ifdef UNO_K2_DEV
    if (getenv("UNO_K2_KS") && atoi(getenv("UNO_K2_KS"))) KS = atoi(getenv("UNO_K2_KS"));
    if (getenv("UNO_K2_REMAP")) adjacent_modes = atoi(getenv("UNO_K2_REMAP"));
#endif
    const int nq = (p.Mc + 15) / 16;
    const int ngroups = (p.N + 4 * NTW - 1) / (4 * NTW), mgroups = (p.M + 4 * MTW - 1) / (4 * MTW);
    const int per_wg = adjacent_modes ? 1 : 4 / KS;
    const int ngw = (ngroups + per_wg - 1) / per_wg, per_group = ngw * mgroups;
    const int sub = 4 / KS;
    const int wg_groups = adjacent_modes ? (p.ncorner * nq + sub - 1) / sub : p.ncorner * nq;      // mode groups (sets of them) dealt to the XCDs
    dim3 grid(((wg_groups + 7) / 8) * 8 * per_group);
    const size_t lds = KS > 1 ? (size_t)4 * MTW * NTW * 8 * 64 * sizeof(float) : 0;
    static int lds_slot[3][64];
    const void* k = p.B.half ? reinterpret_cast<const void*>(mode_gemm_blocks_kernel<MTW, NTW, PF, true>)
                  : p.accumulate ? reinterpret_cast<const void*>(mode_gemm_blocks_kernel<MTW, NTW, PF, false, true>)
                                 : reinterpret_cast<const void*>(mode_gemm_blocks_kernel<MTW, NTW, PF, false>);
    ensure_dynamic_lds(k, lds, lds_slot[p.B.half ? 0 : p.accumulate ? 1 : 2]);
    if (p.B.half) hipLaunchKernelGGL((mode_gemm_blocks_kernel<MTW, NTW, PF, true>), grid, dim3(256), lds, s, p, KS, ngw, per_group, adjacent_modes);
    else if (p.accumulate) hipLaunchKernelGGL((mode_gemm_blocks_kernel<MTW, NTW, PF, false, true>), grid, dim3(256), lds, s, p, KS, ngw, per_group, adjacent_modes);
    else hipLaunchKernelGGL((mode_gemm_blocks_kernel<MTW, NTW, PF, false>), grid, dim3(256), lds, s, p, KS, ngw, per_group, adjacent_modes);
}

// 16 modes: one staging buffer (33.8 KB, also holds the 34.8 KB output tile); 8 modes: two (34.8 KB)
static size_t mode_gemm_lds(int qc, bool pipe) {
    const size_t sb = (size_t)2 * 2 * qc * (KC * 16 + 64 / qc), out = (size_t)16 * 16 * (qc + 1) * 2;
    return std::max((pipe ? 2 : 1) * sb, out) * sizeof(float);
}

// what launch_mode_gemm decides for the 4x4x1 form (its comments there): not used when the last mode group is mostly padding or the
// output of a short-K call is huge; variant 0: <4,4,2> (short K: the weight gradient), 1: <4,2,4> (more than 8 rows), 2: <2,4,4>
struct BlocksPlan { bool use; int variant, KS, adjacent; };
static BlocksPlan blocks_plan(const ModeGemmParams& p) {
    const int groups = (p.Mc + 15) / 16;
    const bool padded = 16 * groups * 5 > p.Mc * 6;
    const bool short_k = p.K <= 32 && p.M >= 16 && p.N >= 16;
    const bool huge_out = short_k && 8.0 * p.M * p.N * p.Mc * p.ncorner > 200e6;
    if (padded || huge_out) return BlocksPlan{false, 0, 1, 0};
    const bool wide_m = p.M > 8;
    const int tm = wide_m ? 16 : 8, tn = wide_m ? 8 : 16;
    const long long tiles_per_group = (long long)((p.M + tm - 1) / tm) * ((p.N + tn - 1) / tn);
    const long long tasks = (long long)p.ncorner * groups * tiles_per_group;
    const bool adjacent = groups >= 48 && (long long)((p.ncorner * groups + 3) / 4) * tiles_per_group >= 200;
    const int KS = (short_k || adjacent) ? 1 : (tasks < 1024 && p.K >= 32) ? 4 : (tasks < 2048 && p.K >= 16) ? 2 : 1;
    return BlocksPlan{true, short_k ? 0 : wide_m ? 1 : 2, KS, adjacent ? 1 : 0};
}
struct BlocksGrid { BlocksCfg cfg; int grid; size_t lds; };
static BlocksGrid blocks_grid(const ModeGemmParams& p, int MTW, int NTW, int KS, int adjacent_modes) {
    const int nq = (p.Mc + 15) / 16;
    const int ngroups = (p.N + 4 * NTW - 1) / (4 * NTW), mgroups = (p.M + 4 * MTW - 1) / (4 * MTW);
    const int per_wg = adjacent_modes ? 1 : 4 / KS;
    const int ngw = (ngroups + per_wg - 1) / per_wg, per_group = ngw * mgroups;
    const int sub = 4 / KS;
    const int wg_groups = adjacent_modes ? (p.ncorner * nq + sub - 1) / sub : p.ncorner * nq;
    return BlocksGrid{BlocksCfg{KS, ngw, per_group, adjacent_modes}, ((wg_groups + 7) / 8) * 8 * per_group,
                      KS > 1 ? (size_t)4 * MTW * NTW * 8 * 64 * sizeof(float) : 0};
}

static int mode_gemm_operands_ok(const ModeGemmParams& p) {
    if (p.ncorner < 1 || p.ncorner > 4 || p.Mc < 1 || p.M < 1 || p.N < 1 || p.K < 1) {
        set_error("mode_gemm: bad sizes M=%d N=%d K=%d corners=%d modes=%d", p.M, p.N, p.K, p.ncorner, p.Mc);
        return -2;
    }
    const long long spanA = ((long long)(p.M - 1) * p.A.s0 + (long long)(p.K - 1) * p.A.s1 + p.Mc) * 8;
    const long long spanB = ((long long)(p.N - 1) * p.B.s1 + (long long)(p.K - 1) * p.B.s0 + p.Mc) * 8;
    if (spanA >= (1LL << 31) || spanB >= (1LL << 31)) { set_error("mode_gemm: an operand spans %lld bytes per corner (limit 2 GiB)", spanA > spanB ? spanA : spanB); return -2; }
    return 0;
}

int launch_mode_gemm(const ModeGemmParams& p, hipStream_t s);

// pa: the input-gradient GEMM, pb: the weight-gradient GEMM of one backward pass (complex64 operands).  One launch where both take the
// 4x4x1 form in the expected variants, else the two launches.
int launch_mode_gemm_pair(const ModeGemmParams& pa, const ModeGemmParams& pb, hipStream_t s) {
    if (int rc = mode_gemm_operands_ok(pa)) return rc;
    if (int rc = mode_gemm_operands_ok(pb)) return rc;
    const BlocksPlan A = blocks_plan(pa), Bp = blocks_plan(pb);
    if (!A.use || !Bp.use || A.variant == 0 || Bp.variant != 0 || pa.B.half || pb.B.half || pa.accumulate) {
        if (int rc = launch_mode_gemm(pb, s)) return rc;
        return launch_mode_gemm(pa, s);
    }
    const BlocksGrid ga = A.variant == 1 ? blocks_grid(pa, 4, 2, A.KS, A.adjacent) : blocks_grid(pa, 2, 4, A.KS, A.adjacent);
    const BlocksGrid gb = blocks_grid(pb, 4, 4, Bp.KS, Bp.adjacent);
    const size_t lds = std::max(ga.lds, gb.lds);
    const double bytes_a = (8.0 * ((double)pa.M * pa.K + (double)pa.M * pa.N) + 8.0 * (double)pa.K * pa.N) * pa.ncorner * pa.Mc;
    const double bytes_b = (8.0 * ((double)pb.M * pb.K + (double)pb.M * pb.N) + 8.0 * (double)pb.K * pb.N) * pb.ncorner * pb.Mc;
    char name[80];
    snprintf(name, sizeof(name), "uno::mode_gemm_blocks_pair_kernel<%s, %s>", A.variant == 1 ? "4, 2, 4" : "2, 4, 4", pb.accumulate ? "true" : "false");
    ProfScope prof(name, bytes_a + bytes_b - 8.0 * (double)pb.K * pb.N * pb.ncorner * pb.Mc, s);       // (gO counted once)
    const dim3 grid(ga.grid + gb.grid);
    static int lds_slot[4][64];
#define UNO_PAIR(MA, NA, ACCB, SLOT)                                                                                              \
    do {                                                                                                                            \
        auto k = mode_gemm_blocks_pair_kernel<MA, NA, 4, ACCB>;                                                                     \
        ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_slot[SLOT]);                                                  \
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, pa, ga.cfg, ga.grid, pb, gb.cfg);                                            \
    } while (0)
    if (A.variant == 1) { if (pb.accumulate) UNO_PAIR(4, 2, true, 0); else UNO_PAIR(4, 2, false, 1); }
    else { if (pb.accumulate) UNO_PAIR(2, 4, true, 2); else UNO_PAIR(2, 4, false, 3); }
#undef UNO_PAIR
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mode_gemm pair launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_mode_gemm(const ModeGemmParams& p, hipStream_t s) {
    if (p.ncorner < 1 || p.ncorner > 4 || p.Mc < 1 || p.M < 1 || p.N < 1 || p.K < 1) {
        set_error("mode_gemm: bad sizes M=%d N=%d K=%d corners=%d modes=%d", p.M, p.N, p.K, p.ncorner, p.Mc);
        return -2;
    }
    // byte offsets inside an operand are 32-bit (raw buffer loads)
    const long long spanA = ((long long)(p.M - 1) * p.A.s0 + (long long)(p.K - 1) * p.A.s1 + p.Mc) * 8;
    const long long spanB = ((long long)(p.N - 1) * p.B.s1 + (long long)(p.K - 1) * p.B.s0 + p.Mc) * 8;
    if (spanA >= (1LL << 31) || spanB >= (1LL << 31)) { set_error("mode_gemm: an operand spans %lld bytes per corner (limit 2 GiB)", spanA > spanB ? spanA : spanB); return -2; }
    const int tiles = ((p.N + 15) / 16) * ((p.M + 15) / 16);
    // under two workgroups per CU and a long K loop: halve the mode chunk (measured, tools/k2bench.py: 256 -> 256 channels x 2 x 64
    // modes 53 -> 36 us, 192 -> 192 x 2 x 36 modes 42 -> 28 us; with K <= 64 the 64-byte runs cost more than the extra workgroups give)
    // (the 8-mode variant walks K two rows per pass: its clamped tail load is only right for even K - odd channel counts take the
    // 16-mode variant)
    const bool narrow = (long long)p.ncorner * ((p.Mc + 15) / 16) * tiles < 512 && p.K >= 96 && (p.K & 1) == 0;
    const int qc = narrow ? 8 : 16;
    const int nq = (p.Mc + qc - 1) / qc;
    dim3 grid(p.ncorner * nq, (p.N + 15) / 16, (p.M + 15) / 16);
    {
        // each operand counted once: A (M x K), B (K x N), out (M x N) complex64 per mode
        const double per_mode = 8.0 * ((double)p.M * p.K + (double)p.M * p.N) + (p.B.half ? 4.0 : 8.0) * (double)p.K * p.N;
        const double k2_bytes = per_mode * p.ncorner * p.Mc;
        // The 4x4x1 form works on whole groups of 16 modes: layers whose last group is mostly padding (2 x 36 modes = 3 groups per
        // corner, a quarter of the third one used) stay on the LDS-staged form with its 8-mode variant; so do weight gradients
        // with very large outputs (measured, tools/k2bench.py: 256 -> 512 channels x 4 x 216 modes, 906 MB: 263 against 332 us).
        const int groups = (p.Mc + 15) / 16;
        const bool padded = 16 * groups * 5 > p.Mc * 6;                         // > 20 % idle blocks
        const bool short_k = p.K <= 32 && p.M >= 16 && p.N >= 16;               // the weight gradient: K = batch
        const bool huge_out = short_k && 8.0 * p.M * p.N * p.Mc * p.ncorner > 200e6;
        if (!padded && !huge_out) {
            // short K and many rows / columns: 16 x 16 outputs per wave, so that each operand row is re-read by N / 16 (M / 16)
            // waves like in the LDS-staged form; few rows (the batch, in the forward /
            // input-gradient forms): all of them in one wave, more columns
            const bool square = short_k;
            const bool wide_m = p.M > 8;
            const int tm = wide_m ? 16 : 8, tn = wide_m ? 8 : 16;
            const long long tiles_per_group = (long long)((p.M + tm - 1) / tm) * ((p.N + tn - 1) / tn);
            const long long tasks = (long long)p.ncorner * groups * tiles_per_group;
            // many mode groups per corner and at least one four-group workgroup per CU: adjacent mode groups, no K split (see the kernel)
            const bool adjacent = groups >= 48 && (long long)((p.ncorner * groups + 3) / 4) * tiles_per_group >= 200;
            const int KS = (square || adjacent) ? 1 : (tasks < 1024 && p.K >= 32) ? 4 : (tasks < 2048 && p.K >= 16) ? 2 : 1;
            char name[64];
            snprintf(name, sizeof(name), "uno::mode_gemm_blocks_kernel<%s, %s, %s>", square ? "4, 4, 2" : wide_m ? "4, 2, 4" : "2, 4, 4", p.B.half ? "true" : "false",
                     (p.accumulate && !p.B.half) ? "true" : "false");
            ProfScope prof(name, k2_bytes, s);
            if (square) launch_blocks_t<4, 4, 2>(p, KS, adjacent ? 1 : 0, s);
            else if (wide_m) launch_blocks_t<4, 2, 4>(p, KS, adjacent ? 1 : 0, s);
            else launch_blocks_t<2, 4, 4>(p, KS, adjacent ? 1 : 0, s);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) { set_error("mode_gemm launch: %s", hipGetErrorString(e)); return -5; }
            return 0;
        }
        // pipelined K loop: measured better with few mode chunks per layer (2 x 36 / 2 x 64 modes: 26 -> 22, 34 -> 29 us) and
        // worse with many (2 x 196 / 2 x 324 modes: 41 -> 50, 58 -> 67 us)
        const bool pipe = narrow && p.ncorner * nq <= 32;
        char name[64];
        snprintf(name, sizeof(name), "uno::mode_gemm_kernel<%d, %s, %s, %s>", qc, pipe ? "true" : "false", p.B.half ? "true" : "false",
                 (p.accumulate && !p.B.half) ? "true" : "false");
        ProfScope prof(name, k2_bytes, s);
        if (p.B.half) {
            if (pipe) hipLaunchKernelGGL((mode_gemm_kernel<8, true, true>), grid, dim3(256), mode_gemm_lds(8, true), s, p);
            else if (narrow) hipLaunchKernelGGL((mode_gemm_kernel<8, false, true>), grid, dim3(256), mode_gemm_lds(8, false), s, p);
            else hipLaunchKernelGGL((mode_gemm_kernel<16, false, true>), grid, dim3(256), mode_gemm_lds(16, false), s, p);
        } else {
            if (p.accumulate) {
                if (pipe) hipLaunchKernelGGL((mode_gemm_kernel<8, true, false, true>), grid, dim3(256), mode_gemm_lds(8, true), s, p);
                else if (narrow) hipLaunchKernelGGL((mode_gemm_kernel<8, false, false, true>), grid, dim3(256), mode_gemm_lds(8, false), s, p);
                else hipLaunchKernelGGL((mode_gemm_kernel<16, false, false, true>), grid, dim3(256), mode_gemm_lds(16, false), s, p);
            } else if (pipe) hipLaunchKernelGGL((mode_gemm_kernel<8, true, false>), grid, dim3(256), mode_gemm_lds(8, true), s, p);
            else if (narrow) hipLaunchKernelGGL((mode_gemm_kernel<8, false, false>), grid, dim3(256), mode_gemm_lds(8, false), s, p);
            else hipLaunchKernelGGL((mode_gemm_kernel<16, false, false>), grid, dim3(256), mode_gemm_lds(16, false), s, p);
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mode_gemm launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
