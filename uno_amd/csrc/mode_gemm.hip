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

namespace uno {

constexpr int KC = 8;           // reduction chunk staged in LDS

// QC = modes per workgroup: 16 (128-byte runs along the mode axis) or 8 (64-byte runs, twice the workgroups: layers with few
// modes and many channels - 2 x 64 modes x 256 x 256 channels - give only 128 workgroups of 16 modes, and a workgroup's phases
// (stage to LDS, issue loads, multiply) do not overlap with one wave per SIMD: the time was the sum of the three)
// PIPE: software-pipelined K loop over two LDS buffers (8-mode variant on layers with few mode chunks, see the launcher)
// BH: operand B (the weights of ops 0 / 1) is stored as half-precision (re, im) pairs - config C5's weight storage - and widened
// as it is loaded; everything after the load is the complex64 kernel.
template <int QC, bool PIPE, bool BH>
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
    for (int e = tid; e < 16 * 16 * QC; e += 256) {
        const int q = e & (QC - 1), mn = e / QC;
        const int m = mn >> 4, n = mn & 15;
        if (q < nmodes && m0 + m < p.M && n0 + n < p.N)
            Ob[(long long)(m0 + m) * p.o_sm + (long long)(n0 + n) * p.o_sn + q] = sO[mn * (QC + 1) + q];
    }
}

// 16 modes: one staging buffer (33.8 KB, also holds the 34.8 KB output tile); 8 modes: two (34.8 KB)
static size_t mode_gemm_lds(int qc, bool pipe) {
    const size_t sb = (size_t)2 * 2 * qc * (KC * 16 + 64 / qc), out = (size_t)16 * 16 * (qc + 1) * 2;
    return std::max((pipe ? 2 : 1) * sb, out) * sizeof(float);
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
    const bool narrow = (long long)p.ncorner * ((p.Mc + 15) / 16) * tiles < 512 && p.K >= 96;
    const int qc = narrow ? 8 : 16;
    const int nq = (p.Mc + qc - 1) / qc;
    dim3 grid(p.ncorner * nq, (p.N + 15) / 16, (p.M + 15) / 16);
    {
        // each operand counted once: A (M x K), B (K x N), out (M x N) complex64 per mode
        const double per_mode = 8.0 * ((double)p.M * p.K + (double)p.M * p.N) + (p.B.half ? 4.0 : 8.0) * (double)p.K * p.N;
        ProfScope prof("uno::mode_gemm_kernel", per_mode * p.ncorner * p.Mc, s);
        // pipelined K loop: measured better with few mode chunks per layer (2 x 36 / 2 x 64 modes: 26 -> 22, 34 -> 29 us) and
        // worse with many (2 x 196 / 2 x 324 modes: 41 -> 50, 58 -> 67 us)
        const bool pipe = narrow && p.ncorner * nq <= 32;
        if (p.B.half) {
            if (pipe) hipLaunchKernelGGL((mode_gemm_kernel<8, true, true>), grid, dim3(256), mode_gemm_lds(8, true), s, p);
            else if (narrow) hipLaunchKernelGGL((mode_gemm_kernel<8, false, true>), grid, dim3(256), mode_gemm_lds(8, false), s, p);
            else hipLaunchKernelGGL((mode_gemm_kernel<16, false, true>), grid, dim3(256), mode_gemm_lds(16, false), s, p);
        } else {
            if (pipe) hipLaunchKernelGGL((mode_gemm_kernel<8, true, false>), grid, dim3(256), mode_gemm_lds(8, true), s, p);
            else if (narrow) hipLaunchKernelGGL((mode_gemm_kernel<8, false, false>), grid, dim3(256), mode_gemm_lds(8, false), s, p);
            else hipLaunchKernelGGL((mode_gemm_kernel<16, false, false>), grid, dim3(256), mode_gemm_lds(16, false), s, p);
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mode_gemm launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
