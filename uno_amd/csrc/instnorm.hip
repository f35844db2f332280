// K13 - InstanceNorm (affine, biased variance, no running statistics) fused with the GELU that follows it in an
// operator block (reference integral_operators.py:269-270, 277-283: `normalize_layer = nn.InstanceNorm2d(out, affine=True)`,
// `x_out = self.normalize_layer(x_out)`, `x_out = F.gelu(x_out)`; 3-D: :497-498, 506-512).
//
// One workgroup per (sample, channel) row of N = prod(grid) contiguous floats.  Statistics are two-pass (mean, then
// sum of squared deviations) like the CPU reference, so results match it to f32 rounding - MIOpen's batch-norm path,
// which torch uses for InstanceNorm on ROCm, is 3e-4 off at odd sizes.  The row is read from HBM once: the later
// sweeps of the same workgroup hit L2 (a row is 49-800 KB).
//   forward : y = [gelu](gamma * (x - mean) * rstd + beta);  saves mean, rstd per row.          HBM: read x, write y
//   backward: g_z = [gelu'(z)] * gy;  S1 = sum g_z, S2 = sum g_z * xhat  (per row; also the bias / weight gradients)
//             gx = gamma * rstd * (g_z - S1 / N - xhat * S2 / N).                               HBM: read x, gy, write gx
#include "uno_common.h"
#include <cstdio>

namespace uno {

__device__ __forceinline__ float in_gelu(float x) { return uno_gelu(x); }
__device__ __forceinline__ float in_dgelu(float x) { return uno_dgelu(x); }

constexpr int IN_T = 512;           // threads per row

// fixed-order block sum: butterfly inside a wave, then the 8 wave sums in order
__device__ __forceinline__ float in_block_sum(float s, float* red) {
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    __syncthreads();                                    // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < IN_T / 64; ++w) t += red[w];
    return t;
}

// Visit the row(s) in 16-byte pieces (rows are only 4-byte aligned): f(k, v0[4], v1[4], n_valid).  Pieces are taken in
// groups of 4 per thread with the NEXT group's loads issued before the current group is consumed; left to the compiler's
// unrolling every group was drained (s_waitcnt vmcnt(0)) before the next was issued - one memory latency per group.
// T = float | unsigned short (bfloat16 bits: widened on load; statistics and arithmetic stay f32)
struct in_f4 { float v[4]; };
template <typename T>
__device__ __forceinline__ in_f4 in_ld4(const T* p) { const float4 t = io_ld4(p); return in_f4{{t.x, t.y, t.z, t.w}}; }

template <bool TWO, typename T, class F>
__device__ __forceinline__ void in_sweep(const T* row0, const T* row1, int N, F f) {
    const int nq = N >> 2;
    if (nq > 0) {
        in_f4 a0[4], a1[4], b0[4], b1[4];
        auto load = [&](int k, in_f4* d0, in_f4* d1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = min((int)threadIdx.x + (k + i) * IN_T, nq - 1);       // clamped: pieces past the end are not consumed
                d0[i] = in_ld4(row0 + 4 * q);
                if (TWO) d1[i] = in_ld4(row1 + 4 * q);
            }
        };
        auto use = [&](int k, const in_f4* d0, const in_f4* d1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = threadIdx.x + (k + i) * IN_T;
                if (q < nq) f(4 * q, d0[i].v, TWO ? d1[i].v : d0[i].v, 4);
            }
        };
        const int cnt = (nq - (int)threadIdx.x + IN_T - 1) / IN_T;                   // pieces of this thread (<= 0: none)
        load(0, a0, a1);
        for (int k = 0; k < cnt; k += 8) {
            load(k + 4, b0, b1);
            __builtin_amdgcn_sched_barrier(0);
            use(k, a0, a1);
            load(k + 8, a0, a1);
            __builtin_amdgcn_sched_barrier(0);
            use(k + 4, b0, b1);
        }
    }
    const int tail = N & 3;
    if (tail && threadIdx.x == 0) {
        float v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < tail; ++i) { v0[i] = io_widen(row0[4 * nq + i]); if (TWO) v1[i] = io_widen(row1[4 * nq + i]); }
        f(4 * nq, v0, v1, tail);
    }
}

template <bool GELU, typename T>
__global__ __launch_bounds__(IN_T) void instnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int C, int N, float eps, int rev) {
    __shared__ float red[IN_T / 64];
    const int r = sweep_x(rev), c = r % C;       // (rev: rows in descending order on every other launch, uno_common.h)
    const T* row = x + (size_t)r * N;
    T* dst = y + (size_t)r * N;
    float s = 0.f;
    in_sweep<false>(row, row, N, [&](int, const float* v, const float*, int n) { for (int i = 0; i < n; ++i) s += v[i]; });
    const float mean = in_block_sum(s, red) / (float)N;
    float q = 0.f;
    in_sweep<false>(row, row, N, [&](int, const float* v, const float*, int n) { for (int i = 0; i < n; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); } });
    const float var = in_block_sum(q, red) / (float)N;
    const float rstd = 1.f / sqrtf(var + eps);
    if (threadIdx.x == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float a = g * rstd, sh = b - mean * a;                    // z = a * x + sh
    in_sweep<false>(row, row, N, [&](int k, const float* v, const float*, int n) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float z = fmaf(a, v[i], sh); o[i] = GELU ? in_gelu(z) : z; }
        if (n == 4) {
            io_store4(dst + k, o[0], o[1], o[2], o[3]);
        } else {
            for (int i = 0; i < n; ++i) io_store1(dst + k + i, o[i]);
        }
    });
}

template <bool GELU, typename T>
__global__ __launch_bounds__(IN_T) void instnorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            T* __restrict__ gx, float* __restrict__ s1_out, float* __restrict__ s2_out, int C, int N, int rev) {
    __shared__ float red[IN_T / 64];
    const int r = sweep_x(rev), c = r % C;       // (rev: rows in descending order on every other launch, uno_common.h)
    const T* row = x + (size_t)r * N;
    const T* grow = gy + (size_t)r * N;
    T* dst = gx + (size_t)r * N;
    const float mean = mean_in[r], rstd = rstd_in[r];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    auto gz_of = [&](float xv, float gv, float& xhat) {
        xhat = (xv - mean) * rstd;
        return GELU ? in_dgelu(fmaf(g, xhat, b)) * gv : gv;
    };
    in_sweep<true>(row, grow, N, [&](int, const float* xv, const float* gv, int n) {
        for (int i = 0; i < n; ++i) { float xh; const float gz = gz_of(xv[i], gv[i], xh); s1 += gz; s2 = fmaf(gz, xh, s2); }
    });
    const float S1 = in_block_sum(s1, red), S2 = in_block_sum(s2, red);
    if (threadIdx.x == 0) { s1_out[r] = S1; s2_out[r] = S2; }
    const float m1 = S1 / (float)N, m2 = S2 / (float)N, gr = g * rstd;
    in_sweep<true>(row, grow, N, [&](int k, const float* xv, const float* gv, int n) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { float xh; const float gz = gz_of(xv[i], gv[i], xh); o[i] = gr * (gz - m1 - xh * m2); }
        if (n == 4) {
            io_store4(dst + k, o[0], o[1], o[2], o[3]);
        } else {
            for (int i = 0; i < n; ++i) io_store1(dst + k + i, o[i]);
        }
    });
}

// ---- register-resident forms.  The sweeps above read a row three times (forward) / twice twice (backward); only the first pass
// is meant to come from HBM, but with every workgroup of the launch resident at once the rows in flight (2048 rows x 199 KB at the
// 128-channel 223^2 level of the Darcy model) are far beyond the L2s, so the later sweeps stream from the Infinity Cache / HBM
// again: 3.4 TB/s on the algorithmic bytes.  A row of up to 512 x 4 NQ floats fits the REGISTERS of its workgroup (NQ <= 32 quads
// per thread forward, 25 backward where x and gy are both held): each element is loaded once, the statistics and the output come
// out of registers.  Same arithmetic and the same fixed-order reductions as the sweep kernels (bit-identical results).
template <bool GELU, typename T, int NQ>
#ifndef UNO_IN_FWD25_WPE
#define UNO_IN_FWD25_WPE 4
#endif
__global__ __launch_bounds__(IN_T, (NQ == 25 ? UNO_IN_FWD25_WPE : 1)) void instnorm_fwd_reg_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, T* __restrict__ y,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, int C, int N, float eps, int rev) {
    __shared__ float red[IN_T / 64];
    const int r = sweep_x(rev), c = r % C;       // (rev: rows in descending order on every other launch, uno_common.h)
    const T* row = x + (size_t)r * N;
    T* dst = y + (size_t)r * N;
    const int nq = N >> 2, tail = N & 3;
    in_f4 v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) v[i] = in_ld4(row + 4 * min((int)threadIdx.x + i * IN_T, max(nq - 1, 0)));
    float tv[3] = {0.f, 0.f, 0.f};
    if (threadIdx.x == 0)
        for (int i = 0; i < tail; ++i) tv[i] = io_widen(row[4 * nq + i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        if ((int)threadIdx.x + i * IN_T < nq) s += ((v[i].v[0] + v[i].v[1]) + v[i].v[2]) + v[i].v[3];
    if (threadIdx.x == 0) for (int i = 0; i < tail; ++i) s += tv[i];
    const float mean = in_block_sum(s, red) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        if ((int)threadIdx.x + i * IN_T < nq) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i].v[e] - mean; q = fmaf(d, d, q); }
        }
    if (threadIdx.x == 0) for (int i = 0; i < tail; ++i) { const float d = tv[i] - mean; q = fmaf(d, d, q); }
    const float var = in_block_sum(q, red) / (float)N;
    const float rstd = 1.f / sqrtf(var + eps);
    if (threadIdx.x == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float a = g * rstd, sh = b - mean * a;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int qi = threadIdx.x + i * IN_T;
        if (qi < nq) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float z = fmaf(a, v[i].v[e], sh); o[e] = GELU ? in_gelu(z) : z; }
            io_store4(dst + 4 * qi, o[0], o[1], o[2], o[3]);
        }
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < tail; ++i) { const float z = fmaf(a, tv[i], sh); io_store1(dst + 4 * nq + i, GELU ? in_gelu(z) : z); }
}

template <bool GELU, typename T, int NQ>
__global__ __launch_bounds__(IN_T) void instnorm_bwd_reg_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                T* __restrict__ gx, float* __restrict__ s1_out, float* __restrict__ s2_out, int C, int N, int rev) {
    __shared__ float red[IN_T / 64];
    const int r = sweep_x(rev), c = r % C;       // (rev: rows in descending order on every other launch, uno_common.h)
    const T* row = x + (size_t)r * N;
    const T* grow = gy + (size_t)r * N;
    T* dst = gx + (size_t)r * N;
    const int nq = N >> 2, tail = N & 3;
    const float mean = mean_in[r], rstd = rstd_in[r];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    // held per element: xhat and g_z (the two sweeps of the other kernel recompute them from x and gy)
    in_f4 xh[NQ], gz[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int qc = min((int)threadIdx.x + i * IN_T, max(nq - 1, 0));
        xh[i] = in_ld4(row + 4 * qc);
        gz[i] = in_ld4(grow + 4 * qc);
    }
    float txh[3] = {0.f, 0.f, 0.f}, tgz[3] = {0.f, 0.f, 0.f};
    if (threadIdx.x == 0)
        for (int i = 0; i < tail; ++i) { txh[i] = io_widen(row[4 * nq + i]); tgz[i] = io_widen(grow[4 * nq + i]); }
    auto to_pair = [&](float& xv, float& gv) {          // (x, gy) -> (xhat, g_z)
        const float h = (xv - mean) * rstd;
        gv = GELU ? in_dgelu(fmaf(g, h, b)) * gv : gv;
        xv = h;
    };
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const bool valid = (int)threadIdx.x + i * IN_T < nq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            to_pair(xh[i].v[e], gz[i].v[e]);
            if (valid) { s1 += gz[i].v[e]; s2 = fmaf(gz[i].v[e], xh[i].v[e], s2); }
        }
        // one quad at a time: left to interleave the erf / exp chains of many quads the compiler needs ~15 temporaries per chain on
        // top of the 2 x 4 NQ held values and spills (337 registers at NQ = 25)
        if (GELU) __builtin_amdgcn_sched_barrier(0);
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < tail; ++i) { to_pair(txh[i], tgz[i]); s1 += tgz[i]; s2 = fmaf(tgz[i], txh[i], s2); }
    const float S1 = in_block_sum(s1, red), S2 = in_block_sum(s2, red);
    if (threadIdx.x == 0) { s1_out[r] = S1; s2_out[r] = S2; }
    const float m1 = S1 / (float)N, m2 = S2 / (float)N, gr = g * rstd;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int qi = threadIdx.x + i * IN_T;
        if (qi < nq)
            io_store4(dst + 4 * qi, gr * (gz[i].v[0] - m1 - xh[i].v[0] * m2), gr * (gz[i].v[1] - m1 - xh[i].v[1] * m2),
                      gr * (gz[i].v[2] - m1 - xh[i].v[2] * m2), gr * (gz[i].v[3] - m1 - xh[i].v[3] * m2));
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < tail; ++i) io_store1(dst + 4 * nq + i, gr * (tgz[i] - m1 - txh[i] * m2));
}

// quads per thread of the register-resident forms for a row of N floats (0: the row is too long - sweep kernels)
static int instnorm_reg_quads(long long N, int max_quads) {
    const long long need = ((N >> 2) + IN_T - 1) / IN_T;
    for (int nq : {4, 8, 13, 16, 25, 32})
        if (need <= nq && nq <= max_quads) return nq;
    return 0;
}

int launch_instnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long long rows, int C,
                        long long N, float eps, int gelu, int bf16, hipStream_t s) {
    typedef unsigned short bf_t;
    if (rows > 0x7fffffffLL || N > 0x7fffffffLL) { set_error("instnorm: too many rows or row too long"); return -2; }
    const int rev = next_sweep_reversed(SWEEP_NORM);
    if (const int rq = (N >= 4 ? instnorm_reg_quads(N, 32) : 0)) {
        ProfScope prof("uno::instnorm_fwd_reg_kernel", (bf16 ? 4.0 : 8.0) * rows * (double)N, s);
        const dim3 grid((unsigned)rows);
#define UNO_IN_FWD(G, TT, Q) hipLaunchKernelGGL((instnorm_fwd_reg_kernel<G, TT, Q>), grid, dim3(IN_T), 0, s, (const TT*)x, gamma, beta, (TT*)y, mean, rstd, C, (int)N, eps, rev)
#define UNO_IN_FWD_Q(G, TT) do { switch (rq) { case 4: UNO_IN_FWD(G, TT, 4); break; case 8: UNO_IN_FWD(G, TT, 8); break; case 13: UNO_IN_FWD(G, TT, 13); break; \
                                   case 16: UNO_IN_FWD(G, TT, 16); break; case 25: UNO_IN_FWD(G, TT, 25); break; default: UNO_IN_FWD(G, TT, 32); } } while (0)
        if (bf16) { if (gelu) UNO_IN_FWD_Q(true, bf_t); else UNO_IN_FWD_Q(false, bf_t); }
        else { if (gelu) UNO_IN_FWD_Q(true, float); else UNO_IN_FWD_Q(false, float); }
#undef UNO_IN_FWD_Q
#undef UNO_IN_FWD
    } else {
        ProfScope prof("uno::instnorm_fwd_kernel", (bf16 ? 4.0 : 8.0) * rows * (double)N, s);
        const dim3 grid((unsigned)rows);
        if (bf16) {
            if (gelu) hipLaunchKernelGGL((instnorm_fwd_kernel<true, bf_t>), grid, dim3(IN_T), 0, s, (const bf_t*)x, gamma, beta, (bf_t*)y, mean, rstd, C, (int)N, eps, rev);
            else hipLaunchKernelGGL((instnorm_fwd_kernel<false, bf_t>), grid, dim3(IN_T), 0, s, (const bf_t*)x, gamma, beta, (bf_t*)y, mean, rstd, C, (int)N, eps, rev);
        } else {
            if (gelu) hipLaunchKernelGGL((instnorm_fwd_kernel<true, float>), grid, dim3(IN_T), 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, C, (int)N, eps, rev);
            else hipLaunchKernelGGL((instnorm_fwd_kernel<false, float>), grid, dim3(IN_T), 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, C, (int)N, eps, rev);
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("instnorm launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_instnorm_bwd(const void* x, const void* gy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        void* gx, float* s1, float* s2, long long rows, int C, long long N, int gelu, int bf16, hipStream_t s) {
    typedef unsigned short bf_t;
    if (rows > 0x7fffffffLL || N > 0x7fffffffLL) { set_error("instnorm: too many rows or row too long"); return -2; }
    const int rev = next_sweep_reversed(SWEEP_NORM);
    // (rounds 3-4 kept the GELU form to 8 quads per thread: with the library erff - a branching piecewise form - the compiler
    // spilled 40 registers at 13 quads and 344 at 25; with the branch-free uno_erf every size fits: 255 registers, no spills, at 25.
    // The 128-channel 223^2 level of the Darcy model - rows of 49 729 floats - left the sweep kernel, which read x and gy twice.)
#ifndef UNO_IN_GELU_Q
#define UNO_IN_GELU_Q 25
#endif
    if (const int rq = (N >= 4 ? instnorm_reg_quads(N, gelu ? UNO_IN_GELU_Q : 25) : 0)) {
        ProfScope prof("uno::instnorm_bwd_reg_kernel", (bf16 ? 6.0 : 12.0) * rows * (double)N, s);
        const dim3 grid((unsigned)rows);
#define UNO_IN_BWD(G, TT, Q) hipLaunchKernelGGL((instnorm_bwd_reg_kernel<G, TT, Q>), grid, dim3(IN_T), 0, s, (const TT*)x, (const TT*)gy, gamma, beta, mean, rstd, (TT*)gx, s1, s2, C, (int)N, rev)
#define UNO_IN_BWD_Q(G, TT) do { switch (rq) { case 4: UNO_IN_BWD(G, TT, 4); break; case 8: UNO_IN_BWD(G, TT, 8); break; case 13: UNO_IN_BWD(G, TT, 13); break; \
                                   case 16: UNO_IN_BWD(G, TT, 16); break; default: UNO_IN_BWD(G, TT, 25); } } while (0)
        if (bf16) { if (gelu) UNO_IN_BWD_Q(true, bf_t); else UNO_IN_BWD_Q(false, bf_t); }
        else { if (gelu) UNO_IN_BWD_Q(true, float); else UNO_IN_BWD_Q(false, float); }
#undef UNO_IN_BWD_Q
#undef UNO_IN_BWD
    } else {
        ProfScope prof("uno::instnorm_bwd_kernel", (bf16 ? 6.0 : 12.0) * rows * (double)N, s);
        const dim3 grid((unsigned)rows);
        if (bf16) {
            if (gelu) hipLaunchKernelGGL((instnorm_bwd_kernel<true, bf_t>), grid, dim3(IN_T), 0, s, (const bf_t*)x, (const bf_t*)gy, gamma, beta, mean, rstd, (bf_t*)gx, s1, s2, C, (int)N, rev);
            else hipLaunchKernelGGL((instnorm_bwd_kernel<false, bf_t>), grid, dim3(IN_T), 0, s, (const bf_t*)x, (const bf_t*)gy, gamma, beta, mean, rstd, (bf_t*)gx, s1, s2, C, (int)N, rev);
        } else {
            if (gelu) hipLaunchKernelGGL((instnorm_bwd_kernel<true, float>), grid, dim3(IN_T), 0, s, (const float*)x, (const float*)gy, gamma, beta, mean, rstd, (float*)gx, s1, s2, C, (int)N, rev);
            else hipLaunchKernelGGL((instnorm_bwd_kernel<false, float>), grid, dim3(IN_T), 0, s, (const float*)x, (const float*)gy, gamma, beta, mean, rstd, (float*)gx, s1, s2, C, (int)N, rev);
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("instnorm backward launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
