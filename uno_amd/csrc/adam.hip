// K10 - one Adam update of one parameter tensor, with the semantics of the reference's optimiser (Adam.py:27-52):
// coupled L2 (g += wd * p) and, for complex parameters, a second moment built from g * conj(g) - one real
// entry per complex entry, shared by its real and imaginary part.  p, g, m are float views (interleaved re/im
// for complex tensors), v has one float per (possibly complex) entry.  One pass: reads p, g, m, v, writes p, m, v.
#include "uno_common.h"

namespace uno {


struct AdamScalars {
    float beta1, beta2, omb1, omb2, eps, wd, step_size, inv_sqrt_bc2;      // omb = 1 - beta, rounded from double
};
struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;            // entries (complex entries for complex tensors)
};

// the update of the 256 x EPT entries starting at entry 256 * EPT * blk of one tensor
template <bool CPLX>
__device__ __forceinline__ void adam_block(const AdamTensor& t, const AdamScalars& a, long long blk) {
    // a thread owns 4 floats of p/g/m = 4 real entries or 2 complex ones
    constexpr int EPT = CPLX ? 2 : 4;
    const long long e0 = (blk * 256 + threadIdx.x) * EPT;
    if (e0 >= t.n) return;
    const long long f0 = CPLX ? 2 * e0 : e0;
    float p[4], g[4], m[4], v[EPT];
    const bool full = e0 + EPT <= t.n;
    const int nf = full ? 4 : (int)((t.n - e0) * (CPLX ? 2 : 1));
    if (full) {
        // 16-byte accesses at 4-byte alignment (gradients may be views at any offset of a flat buffer)
        const f4u p4 = *reinterpret_cast<const f4u*>(t.p + f0), g4 = *reinterpret_cast<const f4u*>(t.g + f0),
                  m4 = *reinterpret_cast<const f4u*>(t.m + f0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[i] = p4.v[i]; g[i] = g4.v[i]; m[i] = m4.v[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[i] = i < nf ? t.p[f0 + i] : 0.f; g[i] = i < nf ? t.g[f0 + i] : 0.f; m[i] = i < nf ? t.m[f0 + i] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) v[i] = e0 + i < t.n ? t.v[e0 + i] : 0.f;
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
        *reinterpret_cast<f4u*>(t.p + f0) = po;
        *reinterpret_cast<f4u*>(t.m + f0) = mo;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nf) { t.p[f0 + i] = p[i]; t.m[f0 + i] = m[i]; }
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i)
        if (e0 + i < t.n) t.v[e0 + i] = v[i];
}

// Multi-tensor form: up to ADAM_MAX_TENSORS parameter tensors per launch, described in the kernel arguments (no device-side
// table to upload).  Workgroup b belongs to the tensor whose range of workgroup indices [first[t], first[t + 1]) holds b: a model's
// dozens of small tensors (biases, 64 x 64 weights) share launches with the large ones instead of one under-filled launch each.
constexpr int ADAM_MAX_TENSORS = 24;
struct AdamMultiParams {
    AdamTensor t[ADAM_MAX_TENSORS];
    unsigned first[ADAM_MAX_TENSORS + 1];
    unsigned cplx_mask;
    int n_tensors;
    AdamScalars a;
    const float* dev_scalars;       // optional: {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), eps, wd} on the device (step count kept on the device)
};

// Step count on the DEVICE (so that the update can be part of a captured HIP graph: the bias corrections then cannot be kernel
// arguments, reference Adam.py:27-52 computes them from state['step'] on the host): one thread advances the counter and evaluates
// the two step-dependent scalars in double, as the host path does.  `hyper` (optional): {lr, eps, weight_decay} as doubles ON THE
// DEVICE - a learning-rate schedule (reference ns_train_2d.py:37,113: StepLR) changes group['lr'] between replays of a captured
// step, and a kernel argument would be frozen into the graph; the caller refreshes the three doubles before a replay instead.
__global__ void adam_advance_kernel(int* step, float* scalars, const double* hyper, double lr, double eps, double wd, double beta1,
                                    double beta2) {
    const int t = *step + 1;
    *step = t;
    if (hyper) { lr = hyper[0]; eps = hyper[1]; wd = hyper[2]; }
    scalars[0] = (float)(lr / (1.0 - pow(beta1, (double)t)));
    scalars[1] = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)t)));
    scalars[2] = (float)eps;
    scalars[3] = (float)wd;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMultiParams q) {
    // (the argument struct itself must stay read-only: written to, it is copied to scratch memory and the descriptor look-up
    // below goes through it - measured 52 ms instead of 4 ms on the NS-3D parameter set)
    AdamScalars a = q.a;
    if (q.dev_scalars) { a.step_size = q.dev_scalars[0]; a.inv_sqrt_bc2 = q.dev_scalars[1]; a.eps = q.dev_scalars[2]; a.wd = q.dev_scalars[3]; }
    int t = 0;
#pragma unroll 1
    for (int i = 1; i < q.n_tensors; ++i)
        if (blockIdx.x >= q.first[i]) t = i;
    const long long blk = blockIdx.x - q.first[t];
    if ((q.cplx_mask >> t) & 1u) adam_block<true>(q.t[t], a, blk);
    else adam_block<false>(q.t[t], a, blk);
}

static AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, double wd, int step) {
    AdamScalars a;
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.wd = (float)wd;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    a.step_size = (float)(lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    return a;
}

int launch_adam_advance(int* step, float* scalars, const double* hyper, double lr, double eps, double wd, double beta1, double beta2,
                        hipStream_t s) {
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, s, step, scalars, hyper, lr, eps, wd, beta1, beta2);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("adam advance launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

int launch_adam_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* n,
                      const int* is_complex, double lr, double beta1, double beta2, double eps, double wd, int step, hipStream_t s,
                      const float* dev_scalars) {
    const AdamScalars a = adam_scalars(lr, beta1, beta2, eps, wd, step);
    int t0 = 0;
    while (t0 < n_tensors) {
        AdamMultiParams q;
        q.a = a; q.cplx_mask = 0; q.n_tensors = 0; q.dev_scalars = dev_scalars;
        unsigned long long blocks = 0;
        double bytes = 0;
        while (t0 < n_tensors && q.n_tensors < ADAM_MAX_TENSORS) {
            const long long nt = n[t0];
            const long long b = (nt + (is_complex[t0] ? 511 : 1023)) / (is_complex[t0] ? 512 : 1024);
            if (blocks + (unsigned long long)b > 0x7fffffffULL) break;
            if (nt > 0) {
                const int k = q.n_tensors++;
                q.t[k] = AdamTensor{p[t0], g[t0], m[t0], v[t0], nt};
                q.first[k] = (unsigned)blocks;
                if (is_complex[t0]) q.cplx_mask |= 1u << k;
                blocks += (unsigned long long)b;
                bytes += (is_complex[t0] ? 8.0 * 5 + 4.0 * 2 : 4.0 * 7) * (double)nt;
            }
            ++t0;
        }
        if (q.n_tensors == 0) {
            if (t0 < n_tensors && blocks == 0) { set_error("adam: tensor %d is too large for one launch", t0); return -2; }
            continue;
        }
        q.first[q.n_tensors] = (unsigned)blocks;
        {
            ProfScope prof("uno::adam_kernel", bytes, s);
            hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, q);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("adam launch: %s", hipGetErrorString(e)); return -5; }
    }
    return 0;
}

int launch_adam(float* p, const float* g, float* m, float* v, long long n, int is_complex, double lr, double beta1, double beta2,
                double eps, double wd, int step, hipStream_t s) {
    return launch_adam_multi(1, &p, &g, &m, &v, &n, &is_complex, lr, beta1, beta2, eps, wd, step, s, nullptr);
}

}  // namespace uno
