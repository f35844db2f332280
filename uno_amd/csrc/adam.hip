// K10 - one Adam update of one parameter tensor, with the semantics of the reference's optimiser (Adam.py:27-52):
// coupled L2 (g += wd * p) and, for complex parameters, a second moment built from g * conj(g) - one real
// entry per complex entry, shared by its real and imaginary part.  p, g, m are float views (interleaved re/im
// for complex tensors), v has one float per (possibly complex) entry.  One pass: reads p, g, m, v, writes p, m, v.
#include "uno_common.h"

namespace uno {

struct AdamParams {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;            // entries (complex entries for complex tensors)
    float beta1, beta2, omb1, omb2, eps, wd, step_size, inv_sqrt_bc2;      // omb = 1 - beta, rounded from double
};

template <bool CPLX>
__global__ __launch_bounds__(256) void adam_kernel(AdamParams a) {
    // a thread owns 4 floats of p/g/m = 4 real entries or 2 complex ones
    constexpr int EPT = CPLX ? 2 : 4;
    const long long e0 = ((long long)blockIdx.x * 256 + threadIdx.x) * EPT;
    if (e0 >= a.n) return;
    const long long f0 = CPLX ? 2 * e0 : e0;
    float p[4], g[4], m[4], v[EPT];
    const bool full = e0 + EPT <= a.n;
    const int nf = full ? 4 : (int)((a.n - e0) * (CPLX ? 2 : 1));
    if (full) {
        // 16-byte accesses at 4-byte alignment (gradients may be views at any offset of a flat buffer)
        const f4u p4 = *reinterpret_cast<const f4u*>(a.p + f0), g4 = *reinterpret_cast<const f4u*>(a.g + f0),
                  m4 = *reinterpret_cast<const f4u*>(a.m + f0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[i] = p4.v[i]; g[i] = g4.v[i]; m[i] = m4.v[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[i] = i < nf ? a.p[f0 + i] : 0.f; g[i] = i < nf ? a.g[f0 + i] : 0.f; m[i] = i < nf ? a.m[f0 + i] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) v[i] = e0 + i < a.n ? a.v[e0 + i] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        g[i] = fmaf(a.wd, p[i], g[i]);
        m[i] = fmaf(a.omb1, g[i], a.beta1 * m[i]);
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const float sq = CPLX ? fmaf(g[2 * i], g[2 * i], g[2 * i + 1] * g[2 * i + 1]) : g[i] * g[i];
        v[i] = fmaf(a.omb2, sq, a.beta2 * v[i]);
        const float denom = sqrtf(v[i]) * a.inv_sqrt_bc2 + a.eps;
        if (CPLX) {
            p[2 * i] -= a.step_size * (m[2 * i] / denom);
            p[2 * i + 1] -= a.step_size * (m[2 * i + 1] / denom);
        } else {
            p[i] -= a.step_size * (m[i] / denom);
        }
    }
    if (full) {
        f4u po, mo;
#pragma unroll
        for (int i = 0; i < 4; ++i) { po.v[i] = p[i]; mo.v[i] = m[i]; }
        *reinterpret_cast<f4u*>(a.p + f0) = po;
        *reinterpret_cast<f4u*>(a.m + f0) = mo;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nf) { a.p[f0 + i] = p[i]; a.m[f0 + i] = m[i]; }
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i)
        if (e0 + i < a.n) a.v[e0 + i] = v[i];
}

int launch_adam(float* p, const float* g, float* m, float* v, long long n, int is_complex, double lr, double beta1, double beta2,
                double eps, double wd, int step, hipStream_t s) {
    AdamParams a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.wd = (float)wd;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    a.step_size = (float)(lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const int ept = is_complex ? 2 : 4;
    const long long threads = (n + ept - 1) / ept;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    {
        ProfScope prof("uno::adam_kernel", (is_complex ? 8.0 * 5 + 4.0 * 2 : 4.0 * 7) * (double)n, s);
        if (is_complex) hipLaunchKernelGGL(adam_kernel<true>, dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(adam_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("adam launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
