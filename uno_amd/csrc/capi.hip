// C ABI of libuno_spectral.so (declared in include/uno_spectral.h) + twiddle-table cache.
#include "../../include/uno_spectral.h"
#include "uno_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace uno {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// (cos, sin)(2 pi n / N) evaluated in double with the phase reduced in integers, exact at the
// multiples of pi/2 (so that sin(pi l) terms of Nyquist / w = 0 columns vanish identically).
float2 twiddle_value(long long n, int N) {
    const double two_pi = 6.283185307179586476925286766559;
    double c, s;
    const long long n4 = 4LL * n;
    if (n4 % N == 0) {
        switch ((n4 / N) & 3) {
            case 0: c = 1; s = 0; break;
            case 1: c = 0; s = 1; break;
            case 2: c = -1; s = 0; break;
            default: c = 0; s = -1; break;
        }
    } else {
        c = std::cos(two_pi * n / N);
        s = std::sin(two_pi * n / N);
    }
    return make_float2((float)c, (float)s);
}
static void fill_twiddles(int N, std::vector<float2>& t) {
    t.resize(N);
    for (int n = 0; n < N; ++n) t[n] = twiddle_value(n, N);
}

void* upload_table(const void* host, size_t bytes) {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) return nullptr;
    void* d = nullptr;
    hipStream_t st = nullptr;
    bool ok = hipMalloc(&d, bytes) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
              hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    if (st) (void)hipStreamDestroy(st);
    if (!ok && d) { (void)hipFree(d); d = nullptr; }
    (void)hipThreadExchangeStreamCaptureMode(&mode);          // back to the caller's mode
    return d;
}

const float2* twiddle_table(int N) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, float2*> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({dev, N});
    if (it != cache.end()) return it->second;
    std::vector<float2> host;
    fill_twiddles(N, host);
    float2* d = static_cast<float2*>(upload_table(host.data(), sizeof(float2) * N));
    if (!d) {
        set_error("twiddle table allocation for N=%d failed: %s", N, hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    cache[{dev, N}] = d;
    return d;
}

// ---------------------------------------------------------------------------- profiling
struct ProfRecord { char name[64]; double bytes; hipEvent_t e0, e1; float ms; };
static std::mutex g_prof_mu;
static std::vector<ProfRecord> g_prof;
static int g_prof_cap = 0;
static bool g_prof_on = false;

ProfScope::ProfScope(const char* name, double bytes, hipStream_t s) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof_on || (int)g_prof.size() >= g_prof_cap) return;
    ProfRecord r;
    snprintf(r.name, sizeof(r.name), "%s", name);
    r.bytes = bytes; r.ms = 0.f;
    if (hipEventCreate(&r.e0) != hipSuccess) return;
    if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
    (void)hipEventRecord(r.e0, s);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[slot].e1, stream);
}

static int check_modes2d(const char* who, int H, int W, int Ho, int Wo, int m1, int m2) {
    if (H < 1 || W < 1 || Ho < 1 || Wo < 1) { set_error("%s: empty grid %dx%d -> %dx%d", who, H, W, Ho, Wo); return -1; }
    if (m1 < 1 || m1 > H || m1 > Ho) {
        set_error("%s: modes1=%d incompatible with grid rows %d -> %d (need 1 <= modes1 <= min rows)", who, m1, H, Ho);
        return -1;
    }
    if (m2 < 1 || m2 > W / 2 + 1 || m2 > Wo / 2 + 1) {
        set_error("%s: modes2=%d incompatible with grid cols %d -> %d (need modes2 <= cols/2+1)", who, m2, W, Wo);
        return -1;
    }
    return 0;
}

// caller-provided scratch of the calling thread (uno_scratch_provide): the any-mode transforms' intermediate spectrum
struct Scratch { void* ptr; size_t bytes; };
static thread_local Scratch t_scratch = {nullptr, 0};

static int dft2d(bool inverse, const float* in, float* out, int n_img, int H, int W, int m1, int m2, float scale,
                 int herm, int mask, hipStream_t s, int sp_group = 0, int sp_stride = 0, int sp_offset = 0, int bf16 = 0) {
    const char* who = inverse ? "uno_dft2d_inverse" : "uno_dft2d_forward";
    if (n_img < 0) { set_error("%s: negative image count", who); return -1; }
    if (n_img > 0 && (!in || !out)) { set_error("%s: null pointer", who); return -1; }
    if (int rc = check_modes2d(who, H, W, H, W, m1, m2)) return rc;
    if (n_img == 0) return 0;
    Dft2dParams p;
    p.in = in; p.out = out; p.n_img = n_img; p.H = H; p.W = W; p.m1 = m1; p.m2 = m2;
    p.scale = scale; p.herm = herm ? 1 : 0; p.mask = mask ? 1 : 0; p.bf16 = bf16 ? 1 : 0; p.rowfreq = nullptr; p.nw = 1; p.exp = 0; p.accumulate = 0; p.act_out = nullptr;
    if (sp_group <= 0) { sp_group = n_img; sp_stride = 0; sp_offset = 0; }          // plain layout: spectrum i of image i
    if (sp_offset < 0 || sp_stride < sp_offset + sp_group || n_img % sp_group) {
        if (!(sp_stride == 0 && sp_offset == 0 && sp_group == n_img)) {
            set_error("%s: bad spectrum grouping (group %d, stride %d, offset %d, images %d)", who, sp_group, sp_stride, sp_offset, n_img);
            return -1;
        }
    }
    p.sp_group = sp_group; p.sp_stride = sp_stride; p.sp_offset = sp_offset;
    p.twH = twiddle_table(H);
    p.twW = twiddle_table(W);
    if (!p.twH || !p.twW) return -6;
    // mode counts beyond the compiled MFMA range (the reference's default modes, integral_operators.py:153-158): any-mode form
    if (m1 > 40 || m2 > 48) return launch_dft2d_generic(p, inverse, t_scratch.ptr, t_scratch.bytes, s);
    // many small images (3-D planes, coarse 2-D levels): plane-batched kernels (dft2d_plane.hip)
    if (inverse ? dft2d_inv_plane_applies(p) : dft2d_fwd_plane_applies(p))
        return inverse ? launch_dft2d_inv_plane(p, s) : launch_dft2d_fwd_plane(p, s);
    // bfloat16 images: row stage on the bf16 MFMA (dft2d_b16.hip)
    if (dft2d_b16_applies(p)) {
        const int rc = inverse ? launch_dft2d_inv_b16(p, s) : launch_dft2d_fwd_b16(p, s);
        if (rc != -3) return rc;        // -3: the shape's LDS need exceeds a CU (tall images with many row modes): the f32-MFMA forms take it
    }
    return inverse ? launch_dft2d_inv(p, s) : launch_dft2d_fwd(p, s);
}

// op 0: forward mix, op 1: grad wrt input spectrum, op 2: weight grad
// rc: 0 = launch p; 1 = nothing to launch (done); negative = error
static int mode_gemm_params(ModeGemmParams& p, int op, const float2* act, const float2* const* w, const float2* go, float2* out_act,
                            float2* const* out_w, int B, int Ci, int Co, int nc, int Mc, hipStream_t s, int w_half, int accumulate) {
    if (B < 0 || Ci < 1 || Co < 1 || nc < 1 || nc > 4 || Mc < 1) {
        set_error("mode gemm: bad sizes B=%d Ci=%d Co=%d corners=%d modes=%d", B, Ci, Co, nc, Mc);
        return -1;
    }
    if (B == 0 && op != 2) return 1;
    const long long P = (long long)nc * Mc;
    p.ncorner = nc; p.Mc = Mc; p.accumulate = (op == 2 && accumulate) ? 1 : 0;
    p.A.half = 0; p.B.half = (op != 2 && w_half) ? 1 : 0;
    for (int c = 0; c < 4; ++c) { p.A.base[c] = nullptr; p.B.base[c] = nullptr; p.out[c] = nullptr; }
    if (op == 0) {              // O[b,o] = sum_i X[b,i] W[i,o]
        p.M = B; p.N = Co; p.K = Ci;
        p.A.s0 = (long long)Ci * P; p.A.s1 = P; p.A.conj = 0;
        p.B.s0 = (long long)Co * Mc; p.B.s1 = Mc; p.B.conj = 0;
        p.o_sm = (long long)Co * P; p.o_sn = P;
        for (int c = 0; c < nc; ++c) { p.A.base[c] = act + (long long)c * Mc; p.B.base[c] = w[c]; p.out[c] = out_act + (long long)c * Mc; }
    } else if (op == 1) {       // gX[b,i] = sum_o gO[b,o] conj(W[i,o])
        p.M = B; p.N = Ci; p.K = Co;
        p.A.s0 = (long long)Co * P; p.A.s1 = P; p.A.conj = 0;
        p.B.s0 = Mc; p.B.s1 = (long long)Co * Mc; p.B.conj = 1;
        p.o_sm = (long long)Ci * P; p.o_sn = P;
        for (int c = 0; c < nc; ++c) { p.A.base[c] = act + (long long)c * Mc; p.B.base[c] = w[c]; p.out[c] = out_act + (long long)c * Mc; }
    } else {                    // gW[i,o] = sum_b conj(X[b,i]) gO[b,o]
        p.M = Ci; p.N = Co; p.K = B;
        p.A.s0 = P; p.A.s1 = (long long)Ci * P; p.A.conj = 1;
        p.B.s0 = (long long)Co * P; p.B.s1 = P; p.B.conj = 0;
        p.o_sm = (long long)Co * Mc; p.o_sn = Mc;
        for (int c = 0; c < nc; ++c) { p.A.base[c] = act + (long long)c * Mc; p.B.base[c] = go + (long long)c * Mc; p.out[c] = out_w[c]; }
        if (B == 0) {
            if (accumulate) return 1;
            for (int c = 0; c < nc; ++c)
                if (hipMemsetAsync(out_w[c], 0, sizeof(float2) * (size_t)Ci * Co * Mc, s) != hipSuccess) { set_error("memset failed"); return -5; }
            return 1;
        }
    }
    return 0;
}

static int mode_gemm(int op, const float2* act, const float2* const* w, const float2* go, float2* out_act,
                     float2* const* out_w, int B, int Ci, int Co, int nc, int Mc, hipStream_t s, int w_half = 0, int accumulate = 0) {
    ModeGemmParams p;
    const int rc = mode_gemm_params(p, op, act, w, go, out_act, out_w, B, Ci, Co, nc, Mc, s, w_half, accumulate);
    if (rc != 0) return rc < 0 ? rc : 0;
    return launch_mode_gemm(p, s);
}

// both GEMMs of a backward pass: gX = gO conj(W) (op 1) and gW (+)= conj(X) gO (op 2), one launch where the kernels allow
static int mode_backward(const float2* xtrunc, const float2* go, const float2* const* w, float2* gx_spec, float2* const* gw, int B, int Ci,
                         int Co, int nc, int Mc, hipStream_t s, int accumulate) {
    ModeGemmParams pa, pb;
    const int ra = mode_gemm_params(pa, 1, go, w, nullptr, gx_spec, nullptr, B, Ci, Co, nc, Mc, s, 0, 0);
    if (ra < 0) return ra;
    const int rb = mode_gemm_params(pb, 2, xtrunc, nullptr, go, nullptr, gw, B, Ci, Co, nc, Mc, s, 0, accumulate);
    if (rb < 0) return rb;
    if (ra == 0 && rb == 0) return launch_mode_gemm_pair(pa, pb, s);
    if (rb == 0) if (int rc = launch_mode_gemm(pb, s)) return rc;
    if (ra == 0) return launch_mode_gemm(pa, s);
    return 0;
}

static int g_reserved_cus = 0;
int reserved_cus() { return __atomic_load_n(&g_reserved_cus, __ATOMIC_RELAXED); }
static int g_sweep_alternation = 255;
static thread_local unsigned t_sweep_count = 0;
// family bits of the setting: 1 K1, 2 K3, 4 K7, 8 K8, 16 K9, 32 InstanceNorm, 64 GELU-projection backward, 128 lift kernels; a launch of a
// family that is switched off runs front to back and does not advance the counter
int next_sweep_reversed(int family) { return (__atomic_load_n(&g_sweep_alternation, __ATOMIC_RELAXED) & family) ? (int)(t_sweep_count++ & 1u) : 0; }

}  // namespace uno

using namespace uno;


// A side stream per device for work that is independent of the caller's critical path (the weight-gradient GEMM of a
// backward call next to input-gradient GEMM + inverse DFT).  fork(): side waits for everything enqueued on `s` so far;
// join(): `s` waits for the side stream.  The mutex is held from fork to join, so concurrent callers on one device take
// turns (enqueueing is short).  Works under stream capture (event fork / join is the capture-safe pattern).
namespace {
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    std::mutex mu;
    bool ok = false;
};
SideStream* side_stream_of_current_device() {
    static std::mutex mu;
    static std::map<int, SideStream*> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    auto it = table.find(dev);
    if (it != table.end()) return it->second->ok ? it->second : nullptr;
    SideStream* ss = new SideStream();
    ss->ok = hipStreamCreateWithFlags(&ss->s, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ss->fork_ev, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ss->join_ev, hipEventDisableTiming) == hipSuccess;
    table[dev] = ss;
    return ss->ok ? ss : nullptr;
}
}  // namespace

extern "C" {

int uno_abi_version(void) { return UNO_SPECTRAL_ABI_VERSION; }

void* uno_upload_table(const void* host, long long bytes) {
    if (!host || bytes < 1) { set_error("uno_upload_table: bad arguments"); return nullptr; }
    void* d = upload_table(host, (size_t)bytes);
    if (!d) set_error("uno_upload_table: allocation / upload of %lld bytes failed: %s", bytes, hipGetErrorString(hipGetLastError()));
    return d;
}

int uno_sweep_alternation(int enable) {
    return __atomic_exchange_n(&uno::g_sweep_alternation, enable == 1 ? 255 : (enable & 255), __ATOMIC_RELAXED);      // (1: all families; other values: a mask, development)
}

int uno_reserve_cus(int n) {
    if (n < 0) n = 0;
    return __atomic_exchange_n(&uno::g_reserved_cus, n, __ATOMIC_RELAXED);
}

int uno_profile_begin(int max_records) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_cap = max_records > 0 ? max_records : 0;
    g_prof.reserve(g_prof_cap);
    g_prof_on = g_prof_cap > 0;
    return 0;
}

int uno_profile_end(void) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof_on = false;
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&r.ms, r.e0, r.e1) != hipSuccess) r.ms = -1.f;
    }
    return (int)g_prof.size();
}

int uno_profile_get(int index, char* name, int name_len, double* ms, double* bytes) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (index < 0 || index >= (int)g_prof.size() || !name || name_len < 1 || !ms || !bytes) {
        set_error("uno_profile_get: bad index or null pointer");
        return -1;
    }
    snprintf(name, (size_t)name_len, "%s", g_prof[index].name);
    *ms = g_prof[index].ms;
    *bytes = g_prof[index].bytes;
    return 0;
}

const char* uno_last_error(void) { return g_err; }

long long uno_dft2d_any_ws_bytes(int n_img, int H, int W, int m1, int m2) {
    (void)W;
    if (n_img <= 0 || H <= 0 || m2 <= 0) return 0;
    return (m1 > 40 || m2 > 48) ? 8LL * n_img * H * m2 : 0;
}

int uno_scratch_provide(void* ptr, long long bytes) {
    if (bytes < 0 || (bytes > 0 && !ptr)) { set_error("uno_scratch_provide: bad buffer"); return -1; }
    t_scratch.ptr = bytes > 0 ? ptr : nullptr;
    t_scratch.bytes = bytes > 0 ? (size_t)bytes : 0;
    return 0;
}

long long uno_spectral_conv2d_fwd_ws_bytes(int B, int Ci, int Co, int m1, int m2) {
    (void)Ci;
    return 8LL * B * Co * 2 * m1 * m2;
}

long long uno_spectral_conv2d_bwd_ws_bytes(int B, int Ci, int Co, int m1, int m2) {
    return 8LL * B * (Ci + Co) * 2 * m1 * m2;
}

int uno_dft2d_forward(const float* images, float* spec, int n_img, int H, int W, int m1, int m2, float scale,
                      int hermitian_cols, int mask_overlap, void* stream) {
    return dft2d(false, images, spec, n_img, H, W, m1, m2, scale, hermitian_cols, mask_overlap, (hipStream_t)stream);
}

int uno_dft2d_inverse(const float* spec, float* images, int n_img, int H, int W, int m1, int m2, float scale,
                      int hermitian_cols, int mask_overlap, void* stream) {
    return dft2d(true, spec, images, n_img, H, W, m1, m2, scale, hermitian_cols, mask_overlap, (hipStream_t)stream);
}

// K3 + up-sampled addend (dft2d_inv_add_kernel.h): where the form applies
int uno_dft2d_inverse_add_applies(int n_img, int H, int W, int m1, int m2, int Hs, int Ws) {
    if (n_img < 1 || H < 1 || W < 1 || m1 < 1 || m2 < 1 || Hs < 1 || Ws < 1 || m1 > H || m2 > W / 2 + 1 || m1 > 40 || m2 > 48) return 0;
    Dft2dParams p;
    p.in = nullptr; p.out = nullptr; p.n_img = n_img; p.H = H; p.W = W; p.m1 = m1; p.m2 = m2; p.scale = 1.f; p.herm = 1; p.mask = 1;
    p.bf16 = 0; p.rowfreq = nullptr; p.nw = 1; p.exp = 0; p.accumulate = 0; p.act_out = nullptr;
    p.sp_group = n_img; p.sp_stride = 0; p.sp_offset = 0; p.twH = nullptr; p.twW = nullptr;
    p.add_Hs = Hs; p.add_Ws = Ws;
    if (dft2d_inv_plane_applies(p)) return 0;           // (many small images take the plane-batched kernels)
    return dft2d_inv_add_applies(p) ? 1 : 0;
}

int uno_dft2d_inverse_add(const float* spec, float* images, int n_img, int H, int W, int m1, int m2, float scale, int hermitian_cols,
                          int mask_overlap, const float* addend, int Hs, int Ws, const int* tile_p0, const float* row_op,
                          const int* col_v0, const float* col_op, void* stream) {
    const char* who = "uno_dft2d_inverse_add";
    if (n_img < 0) { set_error("%s: negative image count", who); return -1; }
    if (n_img > 0 && (!spec || !images || !addend || !tile_p0 || !row_op || !col_v0 || !col_op)) { set_error("%s: null pointer", who); return -1; }
    if (int rc = check_modes2d(who, H, W, H, W, m1, m2)) return rc;
    if (Hs < 1 || Ws < 12 || (long long)Hs * Ws * 4 > 0x7fffffffLL) { set_error("%s: bad addend grid %dx%d", who, Hs, Ws); return -1; }
    if (n_img == 0) return 0;
    if (!uno_dft2d_inverse_add_applies(n_img, H, W, m1, m2, Hs, Ws)) {
        set_error("%s: the fused form does not apply to %d images of %dx%d, modes (%d, %d) (query uno_dft2d_inverse_add_applies)", who, n_img, H, W, m1, m2);
        return -3;
    }
    Dft2dParams p;
    p.in = spec; p.out = images; p.n_img = n_img; p.H = H; p.W = W; p.m1 = m1; p.m2 = m2;
    p.scale = scale; p.herm = hermitian_cols ? 1 : 0; p.mask = mask_overlap ? 1 : 0; p.bf16 = 0; p.rowfreq = nullptr; p.nw = 1; p.exp = 0;
    p.accumulate = 0; p.act_out = nullptr;
    p.sp_group = n_img; p.sp_stride = 0; p.sp_offset = 0;
    p.twH = twiddle_table(H);
    p.twW = twiddle_table(W);
    if (!p.twH || !p.twW) return -6;
    p.add_src = addend; p.add_Hs = Hs; p.add_Ws = Ws; p.add_p0 = tile_p0; p.add_rowop = row_op; p.add_v0 = col_v0; p.add_colop = col_op;
#ifdef UNO_K3A_DEV       // development builds: knock-out switches of the kernel (tools/dev/k3a_time.py), see dft2d_inv_add_kernel.h
    { static const int dev_exp = getenv("UNO_K3A_STAGGER") ? atoi(getenv("UNO_K3A_STAGGER")) : 0; p.exp = dev_exp; }
#endif
    return launch_dft2d_inv_add(p, (hipStream_t)stream);
}

int uno_dft2d_forward_bf16(const void* images, float* spec, int n_img, int H, int W, int m1, int m2, float scale,
                           int hermitian_cols, int mask_overlap, void* stream) {
    return dft2d(false, static_cast<const float*>(images), spec, n_img, H, W, m1, m2, scale, hermitian_cols, mask_overlap,
                 (hipStream_t)stream, 0, 0, 0, 1);
}

int uno_dft2d_inverse_bf16(const float* spec, void* images, int n_img, int H, int W, int m1, int m2, float scale,
                           int hermitian_cols, int mask_overlap, void* stream) {
    return dft2d(true, spec, static_cast<float*>(images), n_img, H, W, m1, m2, scale, hermitian_cols, mask_overlap,
                 (hipStream_t)stream, 0, 0, 0, 1);
}

int uno_dft2d_forward_grouped(const float* images, float* spec, int n_img, int H, int W, int m1, int m2, float scale,
                              int hermitian_cols, int mask_overlap, int group, int stride, int offset, void* stream) {
    if (group < 1) { set_error("uno_dft2d_forward_grouped: group must be positive"); return -1; }
    return dft2d(false, images, spec, n_img, H, W, m1, m2, scale, hermitian_cols, mask_overlap, (hipStream_t)stream, group, stride, offset);
}

int uno_dft2d_forward_grouped_bf16(const void* images, float* spec, int n_img, int H, int W, int m1, int m2, float scale,
                                   int hermitian_cols, int mask_overlap, int group, int stride, int offset, void* stream) {
    if (group < 1) { set_error("uno_dft2d_forward_grouped_bf16: group must be positive"); return -1; }
    return dft2d(false, static_cast<const float*>(images), spec, n_img, H, W, m1, m2, scale, hermitian_cols, mask_overlap,
                 (hipStream_t)stream, group, stride, offset, 1);
}

int uno_dft2d_inverse_grouped_bf16(const float* spec, void* images, int n_img, int H, int W, int m1, int m2, float scale,
                                   int hermitian_cols, int mask_overlap, int group, int stride, int offset, void* stream) {
    if (group < 1) { set_error("uno_dft2d_inverse_grouped_bf16: group must be positive"); return -1; }
    return dft2d(true, spec, static_cast<float*>(images), n_img, H, W, m1, m2, scale, hermitian_cols, mask_overlap,
                 (hipStream_t)stream, group, stride, offset, 1);
}

int uno_dft2d_inverse_grouped(const float* spec, float* images, int n_img, int H, int W, int m1, int m2, float scale,
                              int hermitian_cols, int mask_overlap, int group, int stride, int offset, void* stream) {
    if (group < 1) { set_error("uno_dft2d_inverse_grouped: group must be positive"); return -1; }
    return dft2d(true, spec, images, n_img, H, W, m1, m2, scale, hermitian_cols, mask_overlap, (hipStream_t)stream, group, stride, offset);
}

static int mode_mix_impl(const float* in, const float* const* w, float* out, int op, int B, int Ci, int Co, int ncorner,
                         int modes_per_corner, void* stream, int w_half) {
    if (!w || (B > 0 && (!in || !out))) { set_error("uno_mode_mix: null pointer"); return -1; }
    if (op != 0 && op != 1) { set_error("uno_mode_mix: op must be 0 or 1"); return -1; }
    if (ncorner < 1 || ncorner > 4) { set_error("uno_mode_mix: ncorner=%d out of range", ncorner); return -1; }
    for (int c = 0; c < ncorner; ++c)
        if (!w[c]) { set_error("uno_mode_mix: null weight pointer %d", c); return -1; }
    return mode_gemm(op, reinterpret_cast<const float2*>(in), reinterpret_cast<const float2* const*>(w), nullptr,
                     reinterpret_cast<float2*>(out), nullptr, B, Ci, Co, ncorner, modes_per_corner, (hipStream_t)stream, w_half);
}

int uno_mode_mix(const float* in, const float* const* w, float* out, int op, int B, int Ci, int Co, int ncorner,
                 int modes_per_corner, void* stream) {
    return mode_mix_impl(in, w, out, op, B, Ci, Co, ncorner, modes_per_corner, stream, 0);
}

int uno_mode_mix_f16w(const float* in, const void* const* w, float* out, int op, int B, int Ci, int Co, int ncorner,
                      int modes_per_corner, void* stream) {
    return mode_mix_impl(in, reinterpret_cast<const float* const*>(w), out, op, B, Ci, Co, ncorner, modes_per_corner, stream, 1);
}

static int mode_wgrad_impl(const float* xtrunc, const float* go, float* const* gw, int B, int Ci, int Co, int ncorner,
                           int modes_per_corner, int accumulate, void* stream) {
    if (!gw || (B > 0 && (!xtrunc || !go))) { set_error("uno_mode_wgrad: null pointer"); return -1; }
    if (ncorner < 1 || ncorner > 4) { set_error("uno_mode_wgrad: ncorner=%d out of range", ncorner); return -1; }
    for (int c = 0; c < ncorner; ++c)
        if (!gw[c]) { set_error("uno_mode_wgrad: null output pointer %d", c); return -1; }
    return mode_gemm(2, reinterpret_cast<const float2*>(xtrunc), nullptr, reinterpret_cast<const float2*>(go), nullptr,
                     reinterpret_cast<float2* const*>(gw), B, Ci, Co, ncorner, modes_per_corner, (hipStream_t)stream, 0, accumulate);
}

int uno_mode_wgrad(const float* xtrunc, const float* go, float* const* gw, int B, int Ci, int Co, int ncorner,
                   int modes_per_corner, void* stream) {
    return mode_wgrad_impl(xtrunc, go, gw, B, Ci, Co, ncorner, modes_per_corner, 0, stream);
}

int uno_mode_backward(const float* xtrunc, const float* go, const float* const* w, float* gx_spec, float* const* gw, int B, int Ci, int Co,
                      int ncorner, int modes_per_corner, int accumulate, void* stream) {
    if (!w || !gw || (B > 0 && (!xtrunc || !go || !gx_spec))) { set_error("uno_mode_backward: null pointer"); return -1; }
    if (ncorner < 1 || ncorner > 4) { set_error("uno_mode_backward: ncorner=%d out of range", ncorner); return -1; }
    for (int c = 0; c < ncorner; ++c)
        if (!w[c] || !gw[c]) { set_error("uno_mode_backward: null weight / gradient pointer %d", c); return -1; }
    return mode_backward(reinterpret_cast<const float2*>(xtrunc), reinterpret_cast<const float2*>(go), reinterpret_cast<const float2* const*>(w),
                         reinterpret_cast<float2*>(gx_spec), reinterpret_cast<float2* const*>(gw), B, Ci, Co, ncorner, modes_per_corner,
                         (hipStream_t)stream, accumulate);
}

int uno_mode_wgrad_acc(const float* xtrunc, const float* go, float* const* gw, int B, int Ci, int Co, int ncorner,
                       int modes_per_corner, int accumulate, void* stream) {
    return mode_wgrad_impl(xtrunc, go, gw, B, Ci, Co, ncorner, modes_per_corner, accumulate, stream);
}

static int resample2d_impl(const void* in, void* out, float* tmp, int n_img, int H, int W, int Ho, int Wo, const int* startH,
                           const float* wtH, int KH, const int* startW, const float* wtW, int KW, const int* tile_p0,
                           const float* tile_w, int NP, int accumulate, int bf16, void* stream) {
    if (n_img < 0 || H < 1 || W < 1 || Ho < 1 || Wo < 1) { set_error("uno_resample2d: bad sizes"); return -1; }
    if (n_img == 0) return 0;
    if (!in || !out || !tmp || !startH || !wtH || !startW || !wtW) { set_error("uno_resample2d: null pointer"); return -1; }
    return launch_resample2d(in, out, tmp, n_img, H, W, Ho, Wo, startH, wtH, KH, startW, wtW, KW, tile_p0, tile_w, NP, accumulate, bf16, (hipStream_t)stream);
}

int uno_resample2d(const float* in, float* out, float* tmp, int n_img, int H, int W, int Ho, int Wo, const int* startH,
                   const float* wtH, int KH, const int* startW, const float* wtW, int KW, const int* tile_p0,
                   const float* tile_w, int NP, int accumulate, void* stream) {
    return resample2d_impl(in, out, tmp, n_img, H, W, Ho, Wo, startH, wtH, KH, startW, wtW, KW, tile_p0, tile_w, NP, accumulate, 0, stream);
}

int uno_resample2d_bf16(const void* in, void* out, float* tmp, int n_img, int H, int W, int Ho, int Wo, const int* startH,
                        const float* wtH, int KH, const int* startW, const float* wtW, int KW, const int* tile_p0,
                        const float* tile_w, int NP, int accumulate, void* stream) {
    return resample2d_impl(in, out, tmp, n_img, H, W, Ho, Wo, startH, wtH, KH, startW, wtW, KW, tile_p0, tile_w, NP, accumulate, 1, stream);
}

static int channel_mix_impl(const void* x, const float* w, const float* bias, void* y, int B, int Ci, int Co, long long P,
                            int transpose_w, int accumulate, int act_in, const void* dgelu_of, int bf16, void* stream) {
    if (B < 0 || Ci < 1 || Co < 1 || P < 0) { set_error("uno_channel_mix: bad sizes B=%d Ci=%d Co=%d P=%lld", B, Ci, Co, P); return -1; }
    if (B == 0 || P == 0) return 0;
    if (!x || !w || !y) { set_error("uno_channel_mix: null pointer"); return -1; }
    return launch_channel_mix(x, w, bias, y, B, Ci, Co, P, transpose_w, accumulate, act_in, dgelu_of, bf16, (hipStream_t)stream,
                              t_scratch.ptr, t_scratch.bytes);
}

long long uno_channel_mix_ws_bytes(int Ci, int Co, long long P, int bf16) {
    if (Ci < 1 || Co < 1 || P < 1) return 0;
    return channel_mix_ws_bytes(Ci, Co, P, bf16);
}

int uno_channel_mix(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co, long long P,
                    int transpose_w, int accumulate, int act_in, const float* dgelu_of, void* stream) {
    return channel_mix_impl(x, w, bias, y, B, Ci, Co, P, transpose_w, accumulate, act_in, dgelu_of, 0, stream);
}

int uno_channel_mix_bf16(const void* x, const float* w, const float* bias, void* y, int B, int Ci, int Co, long long P,
                         int transpose_w, int accumulate, int act_in, const void* dgelu_of, void* stream) {
    return channel_mix_impl(x, w, bias, y, B, Ci, Co, P, transpose_w, accumulate, act_in, dgelu_of, 1, stream);
}

static int channel_mix2_impl(const void* x1, const void* x2, int C1, const float* w, const float* bias, void* y1, void* y2, int Co1,
                             void* y_act, int B, int Ci, int Co, long long P, int transpose_w, int accumulate, int act_in,
                             const void* dgelu_of, const float* proj_w, const float* proj_b, void* proj_out, int bf16, void* stream,
                             const PixelWindow& win = PixelWindow()) {
    if (B < 0 || Ci < 1 || Co < 1 || P < 0) { set_error("uno_channel_mix2: bad sizes B=%d Ci=%d Co=%d P=%lld", B, Ci, Co, P); return -1; }
    if (B == 0 || P == 0) return 0;
    if (!x1 || !w || !y1) { set_error("uno_channel_mix2: null pointer"); return -1; }
    ChannelMixArgs a{};
    a.x = x1; a.x2 = x2; a.w = w; a.bias = bias; a.y = y1; a.y2 = y2; a.y_act = y_act; a.dgelu_of = dgelu_of;
    a.B = B; a.Ci = Ci; a.Co = Co; a.C1 = x2 ? C1 : Ci; a.Co1 = y2 ? Co1 : Co; a.P = P;
    a.transpose_w = transpose_w; a.accumulate = accumulate; a.act_in = act_in; a.bf16 = bf16;
    a.proj_w = proj_w; a.proj_b = proj_b; a.proj_out = proj_out;
    a.win = win;
    a.ws = t_scratch.ptr; a.ws_bytes = t_scratch.bytes;
    return launch_channel_mix2(a, (hipStream_t)stream);
}

int uno_channel_mix2(const float* x1, const float* x2, int C1, const float* w, const float* bias, float* y1, float* y2, int Co1,
                     float* y_act, int B, int Ci, int Co, long long P, int transpose_w, int accumulate, int act_in,
                     const float* dgelu_of, const float* proj_w, const float* proj_b, float* proj_out, void* stream) {
    return channel_mix2_impl(x1, x2, C1, w, bias, y1, y2, Co1, y_act, B, Ci, Co, P, transpose_w, accumulate, act_in, dgelu_of,
                             proj_w, proj_b, proj_out, 0, stream);
}

int uno_channel_mix2_bf16(const void* x1, const void* x2, int C1, const float* w, const float* bias, void* y1, void* y2, int Co1,
                          void* y_act, int B, int Ci, int Co, long long P, int transpose_w, int accumulate, int act_in,
                          const void* dgelu_of, const float* proj_w, const float* proj_b, void* proj_out, void* stream) {
    return channel_mix2_impl(x1, x2, C1, w, bias, y1, y2, Co1, y_act, B, Ci, Co, P, transpose_w, accumulate, act_in, dgelu_of,
                             proj_w, proj_b, proj_out, 1, stream);
}

// the pixel axis of a *_win call: rows x cols logical pixels, row r at r * pitch of a channel plane, planes `plane` elements apart
static bool make_window(const char* who, int rows, int cols, int pitch, long long plane, PixelWindow* win, long long* P) {
    if (rows < 1 || cols < 1 || pitch < cols || plane < 1) { set_error("%s: bad window rows=%d cols=%d pitch=%d plane=%lld", who, rows, cols, pitch, plane); return false; }
    win->plane = plane; win->cols = cols; win->pitch = pitch;
    *P = (long long)rows * cols;
    return true;
}

int uno_channel_mix2_win(const float* x1, const float* x2, int C1, const float* w, const float* bias, float* y1, float* y2, int Co1,
                         float* y_act, int B, int Ci, int Co, int rows, int cols, int pitch, long long plane, int transpose_w,
                         int accumulate, int act_in, const float* dgelu_of, const float* proj_w, const float* proj_b, float* proj_out,
                         void* stream) {
    PixelWindow win; long long P;
    if (!make_window("uno_channel_mix2_win", rows, cols, pitch, plane, &win, &P)) return -1;
    return channel_mix2_impl(x1, x2, C1, w, bias, y1, y2, Co1, y_act, B, Ci, Co, P, transpose_w, accumulate, act_in, dgelu_of,
                             proj_w, proj_b, proj_out, 0, stream, win);
}

// ---- the backward pass of `fc2(F.gelu(fc1(cat)))` without the gradient at fc1's output in memory (ABI 12)
static bool project_backward_geometry(const char* who, int rows, int cols, int pitch, long long plane, PixelWindow* win, long long* P) {
    if (rows == 0 && cols == 0 && pitch == 0) {          // dense planes
        if (plane < 1) { if (who) set_error("%s: bad plane size %lld", who, plane); return false; }
        *win = PixelWindow(); *P = plane;
        return true;
    }
    if (rows < 1 || cols < 1 || pitch < cols || plane < 1) { if (who) set_error("%s: bad window rows=%d cols=%d pitch=%d plane=%lld", who, rows, cols, pitch, plane); return false; }
    win->plane = plane; win->cols = cols; win->pitch = pitch;
    *P = (long long)rows * cols;
    if (const char* why = pix_window_error(*win, *P)) { if (who) set_error("%s: %s", who, why); return false; }
    return true;
}

int uno_project_backward_applies(int B, int C1, int Ci, int Co, int rows, int cols, int pitch, long long plane) {
    PixelWindow win; long long P;
    if (B < 1 || B > 65535 || Ci < 1 || Co < 1 || C1 < 1 || C1 > Ci || !project_backward_geometry(nullptr, rows, cols, pitch, plane, &win, &P)) return 0;
    // the input gradients: the wide kernel's general form on fc1's Co channels -> Ci gradient channels, destinations split at C1
    if (Ci % 128 || Co % 16 || Co >= 128 || P < 128 || P % 4 || (C1 < Ci && C1 % 64)) return 0;
    if ((long long)Ci * plane >= (1LL << 29) || (long long)Ci * Co >= (1LL << 30)) return 0;
    return channel_wgrad_pb_applies(B, Ci, Co, C1, P) ? 1 : 0;
}

long long uno_project_backward_ws_bytes(int B, int Ci, int Co, long long P) {
    if (B < 1 || Ci < 1 || Co < 1 || P < 1) return 0;
    return 4LL * channel_wgrad_pb_ws_floats(B, Ci, Co, P);
}

int uno_project_backward(const float* x1, const float* x2, int C1, const float* w, const float* pre, const float* w2, const float* gout,
                         float* g1, float* g2, float* gw, float* gb, float* gw2, float* gb2, void* ws, int B, int Ci, int Co, int rows,
                         int cols, int pitch, long long plane, int act_in, int accumulate_w, void* stream) {
    PixelWindow win; long long P;
    if (B < 0 || Ci < 1 || Co < 1) { set_error("uno_project_backward: bad sizes B=%d Ci=%d Co=%d", B, Ci, Co); return -1; }
    if (!project_backward_geometry("uno_project_backward", rows, cols, pitch, plane, &win, &P)) return -1;
    if (accumulate_w != 0 && accumulate_w != 1) { set_error("uno_project_backward: accumulate_w is 0 or 1"); return -1; }
    if (!gw || !gw2) { set_error("uno_project_backward: null pointer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        if ((!accumulate_w && (hipMemsetAsync(gw, 0, sizeof(float) * Co * Ci, s) != hipSuccess || (gb && hipMemsetAsync(gb, 0, sizeof(float) * Co, s) != hipSuccess))) ||
            hipMemsetAsync(gw2, 0, sizeof(float) * Co, s) != hipSuccess || (gb2 && hipMemsetAsync(gb2, 0, sizeof(float), s) != hipSuccess)) {
            set_error("uno_project_backward: memset failed");
            return -5;
        }
        return 0;
    }
    if (!x2) C1 = Ci;
    if (!uno_project_backward_applies(B, C1, Ci, Co, rows, cols, pitch, plane)) {
        set_error("uno_project_backward: shape outside the fused kernels' range (query uno_project_backward_applies)");
        return -3;
    }
    if (!x1 || !w || !pre || !w2 || !gout || !g1 || (x2 && !g2) || !ws) { set_error("uno_project_backward: null pointer"); return -1; }
    {   // both input gradients from one pass over the pre-activation
        ChannelMixArgs a{};
        a.x = pre; a.w = w; a.y = g1; a.y2 = x2 ? g2 : nullptr; a.dgelu_of = act_in ? x1 : nullptr;
        a.B = B; a.Ci = Co; a.Co = Ci; a.C1 = Co; a.Co1 = x2 ? C1 : Ci; a.P = P; a.transpose_w = 1;
        a.win = win; a.pb_w2 = w2; a.pb_g = gout;
        if (int rc = launch_channel_mix2(a, s)) return rc;
    }
    WgradProjectedBack pb;
    pb.w2 = w2; pb.g = gout; pb.gw2 = gw2; pb.gb2 = gb2;
    return launch_channel_wgrad2(pre, x1, x2, C1, gw, gb, (float*)ws, B, Ci, Co, P, act_in, accumulate_w, 0, s, win, pb);
}

int uno_clear_border(float* t, long long n_planes, int Hp, int Wp, int rows, int cols, void* stream) {
    if (n_planes < 0 || Hp < 1 || Wp < 1 || rows < 0 || rows > Hp || cols < 0 || cols > Wp) {
        set_error("uno_clear_border: bad sizes planes=%lld (%d, %d) keep (%d, %d)", n_planes, Hp, Wp, rows, cols);
        return -1;
    }
    if (n_planes == 0) return 0;
    if (!t) { set_error("uno_clear_border: null pointer"); return -1; }
    return launch_clear_border(t, n_planes, Hp, Wp, rows, cols, (hipStream_t)stream);
}

static int lift_padded(const char* who, const float* x, const float* w, const float* bias, float* y, float* y_act, const float* gmul, int B,
                       int Ci, int Co, int H, int W, int Hp, int Wp, int act_in, void* stream) {
    if (B < 0 || Ci < 1 || Co < 1 || H < 1 || W < 1 || Hp < H || Wp < W) {
        set_error("%s: bad sizes B=%d Ci=%d Co=%d (%d, %d) -> (%d, %d)", who, B, Ci, Co, H, W, Hp, Wp);
        return -1;
    }
    if (B == 0) return 0;
    if (!x || !w || (!y && !y_act) || (gmul && !y)) { set_error("%s: null pointer", who); return -1; }
    ChannelMixArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.y_act = y_act; a.gmul = gmul;
    a.B = B; a.Ci = Ci; a.Co = Co; a.C1 = Ci; a.Co1 = Co; a.P = (long long)H * W; a.act_in = act_in;
    a.act_cols = W; a.act_pitch = Wp; a.act_plane = (long long)Hp * Wp;
    if (int rc = launch_channel_mix2(a, (hipStream_t)stream)) return rc;
    return y_act ? launch_clear_border(y_act, (long long)B * Co, Hp, Wp, H, W, (hipStream_t)stream) : 0;
}

int uno_channel_mix_act_padded(const float* x, const float* w, const float* bias, float* y, float* y_act, int B, int Ci, int Co, int H, int W,
                               int Hp, int Wp, int act_in, void* stream) {
    if (!y_act) { set_error("uno_channel_mix_act_padded: null pointer"); return -1; }
    return lift_padded("uno_channel_mix_act_padded", x, w, bias, y, y_act, nullptr, B, Ci, Co, H, W, Hp, Wp, act_in, stream);
}

int uno_channel_mix_dgelu_padded(const float* x, const float* w, const float* bias, const float* g_padded, float* gz, int B, int Ci, int Co,
                                 int H, int W, int Hp, int Wp, int act_in, void* stream) {
    if (!g_padded || !gz) { set_error("uno_channel_mix_dgelu_padded: null pointer"); return -1; }
    return lift_padded("uno_channel_mix_dgelu_padded", x, w, bias, gz, nullptr, g_padded, B, Ci, Co, H, W, Hp, Wp, act_in, stream);
}

// ---- the whole lift (reference darcy_flow_uno2d.py:98-107) with its first layer's output never stored
static int lift_check(const char* who, int B, int Cin, int Cm, int Co, int H, int W, int Hp, int Wp) {
    if (B < 0 || Cin < 1 || Cin > 3 || Cm < 5 || Cm > 32 || Cm % 16 || Co < 1 || H < 1 || W < 260 || Hp < H || Wp < W || (long long)H * W >= (1LL << 24)) {
        set_error("%s: needs 1 .. 3 input channels, 16 or 32 middle channels, 260 <= W <= Wp, H <= Hp, H * W < 2^24 (got B=%d %d -> %d -> %d, (%d, %d) -> (%d, %d))",
                  who, B, Cin, Cm, Co, H, W, Hp, Wp);
        return -1;
    }
    return 0;
}

// size limits of the fused lift kernels (lift_bwd.hip: 32-bit element offsets into the padded planes, 24-bit pixel slots over H x Wp):
// grids beyond them take the layer-by-layer forms below instead of failing (advisor finding, round 5)
static bool lift_fused_fits(int B, int H, int Hp, int Wp, int Co) {
    return (long long)Hp * Wp * Co < (1LL << 31) && B <= 65535 && (long long)H * Wp < (1LL << 24);
}

int uno_lift_forward(const float* x, const float* w1, const float* b1, const float* w0, const float* b0, float* act, int B, int Cin, int Cm,
                     int Co, int H, int W, int Hp, int Wp, void* stream) {
    if (int rc = lift_check("uno_lift_forward", B, Cin, Cm, Co, H, W, Hp, Wp)) return rc;
    if (B == 0) return 0;
    if (!x || !w1 || !w0 || !act) { set_error("uno_lift_forward: null pointer"); return -1; }
    if (lift_bwd_fused_applies(Cin, Cm, Co, W, (long long)H * W) && lift_fused_fits(B, H, Hp, Wp, Co) && (Wp & ~3) >= 260 && (Wp & ~3) >= W) {      // K16 (lift_bwd.hip): the dedicated kernel at the Darcy widths
        if (int rc = launch_lift_forward_fused(x, w1, b1, w0, b0, act, B, Cin, H, W, Hp, Wp, (hipStream_t)stream)) return rc;
        return launch_clear_border(act, (long long)B * Co, Hp, Wp, H, Wp, (hipStream_t)stream);         // the rows below the domain
    }
    ChannelMixArgs a{};
    a.x = x; a.w = w0; a.bias = b0; a.y = nullptr; a.y_act = act;
    a.B = B; a.Ci = Cm; a.Co = Co; a.C1 = Cm; a.Co1 = Co; a.P = (long long)H * W; a.act_in = 1;
    a.act_cols = W; a.act_pitch = Wp; a.act_plane = (long long)Hp * Wp;
    a.vh_x = x; a.vh_w = w1; a.vh_b = b1; a.vh_ci = Cin; a.vh_mode = 1;
    if (int rc = launch_channel_mix2(a, (hipStream_t)stream)) return rc;
    return launch_clear_border(act, (long long)B * Co, Hp, Wp, H, W, (hipStream_t)stream);
}

// scratch of uno_lift_backward: gz (B, Co, H, W), g_h (B, Cm, H, W), then the larger of the two weight-gradient scratches
static long long lift_wgrad_ws(int B, int Cin, int Cm, int Co, long long P) {
    const long long a = 4LL * channel_wgrad_ws_floats(B, Cm, Co, P, nullptr), b = 4LL * channel_wgrad_ws_floats(B, Cin, Cm, P, nullptr);
    return a > b ? a : b;
}
long long uno_lift_bwd_ws_bytes(int B, int Cin, int Cm, int Co, int H, int W) {
    if (B < 1 || Cin < 1 || Cm < 1 || Co < 1 || H < 1 || W < 1) return 0;
    const long long P = (long long)H * W;
    return 4LL * B * P * (Co + Cm) + lift_wgrad_ws(B, Cin, Cm, Co, P);
}

int uno_lift_backward_takes_second(int B, int Cin, int Cm, int Co, int H, int W, int Hp, int Wp) {
    if (B < 1 || Cin < 1 || Cin > 3 || Cm < 5 || Cm > 32 || Cm % 16 || Co < 1 || H < 1 || W < 260 || Hp < H || Wp < W) return 0;
    return (lift_bwd_fused_applies(Cin, Cm, Co, W, (long long)H * W) && lift_fused_fits(B, H, Hp, Wp, Co)) ? 1 : 0;
}

int uno_lift_backward2(const float* x, const float* w1, const float* b1, const float* w0, const float* b0_, const float* g_act, const float* g_act2,
                       float* gw1, float* gb1, float* gw0, float* gb0, void* ws, int B, int Cin, int Cm, int Co, int H, int W, int Hp, int Wp,
                       void* stream);

int uno_lift_backward(const float* x, const float* w1, const float* b1, const float* w0, const float* b0_, const float* g_act, float* gw1,
                      float* gb1, float* gw0, float* gb0, void* ws, int B, int Cin, int Cm, int Co, int H, int W, int Hp, int Wp, void* stream) {
    return uno_lift_backward2(x, w1, b1, w0, b0_, g_act, nullptr, gw1, gb1, gw0, gb0, ws, B, Cin, Cm, Co, H, W, Hp, Wp, stream);
}

int uno_lift_backward2(const float* x, const float* w1, const float* b1, const float* w0, const float* b0_, const float* g_act, const float* g_act2,
                       float* gw1, float* gb1, float* gw0, float* gb0, void* ws, int B, int Cin, int Cm, int Co, int H, int W, int Hp, int Wp,
                       void* stream) {
    if (int rc = lift_check("uno_lift_backward", B, Cin, Cm, Co, H, W, Hp, Wp)) return rc;
    if (g_act2 && !uno_lift_backward_takes_second(B, Cin, Cm, Co, H, W, Hp, Wp) && B > 0) {
        set_error("uno_lift_backward2: a second gradient tensor goes with the fused kernel only (query uno_lift_backward_takes_second)");
        return -3;
    }
    if (!gw1 || !gw0) { set_error("uno_lift_backward: null pointer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        if (hipMemsetAsync(gw1, 0, sizeof(float) * Cm * Cin, s) != hipSuccess || hipMemsetAsync(gw0, 0, sizeof(float) * Co * Cm, s) != hipSuccess ||
            (gb1 && hipMemsetAsync(gb1, 0, sizeof(float) * Cm, s) != hipSuccess) || (gb0 && hipMemsetAsync(gb0, 0, sizeof(float) * Co, s) != hipSuccess)) {
            set_error("uno_lift_backward: memset failed");
            return -5;
        }
        return 0;
    }
    if (!x || !w1 || !w0 || !g_act || !ws) { set_error("uno_lift_backward: null pointer"); return -1; }
    const long long P = (long long)H * W;
    if (lift_bwd_fused_applies(Cin, Cm, Co, W, P) && lift_fused_fits(B, H, Hp, Wp, Co)) {
        // one kernel per pixel tile: neither gz nor gh leaves the chip (lift_bwd.hip); ws = the two arrays of partial-sum blocks
        float* part = static_cast<float*>(ws);
        const long long nparts = lift_bwd_fused_parts(B, H, W);
        float* part1 = part + (size_t)nparts * Co * (Cm + 1);
        if (int rc = launch_lift_backward_fused(x, w1, b1, w0, b0_, g_act, part, part1, B, Cin, H, W, Hp, Wp, s, g_act2)) return rc;
        if (int rc = launch_channel_wgrad_finish(part, gw0, gb0, Cm, Co, nparts, 0, s)) return rc;
        return launch_channel_wgrad_finish(part1, gw1, gb1, Cin, Cm, nparts, 0, s);
    }
    // (measured and dropped, round 5: batch entries in groups whose gz stays in the 256 MB Infinity Cache between the kernel that writes
    // it and the two that read it - groups of 2 / 4 / 8 of 16 ran the step at 13.1-13.3 / 12.8 / 12.65 ms against 12.37-12.40 whole: the
    // shorter launches lose more to ramp and tail than the cache gives)
    const int G = B;
    float* gz = static_cast<float*>(ws);
    float* gh = gz + (size_t)B * Co * P;
    float* wws = gh + (size_t)B * Cm * P;
    for (int b0 = 0; b0 < B; b0 += G) {
        const int nb = (B - b0 < G) ? B - b0 : G;
        const float* xg = x + (size_t)b0 * Cin * P;
        const float* gg = g_act + (size_t)b0 * Co * Hp * Wp;
        const int acc = b0 > 0 ? 1 : 0;
        // 1. gz = gelu'(fc0(gelu(h))) * g_act[..., :H, :W], the layer recomputed from the virtual h = fc_n1(x)
        {
            ChannelMixArgs a{};
            a.x = xg; a.w = w0; a.bias = b0_; a.y = gz; a.gmul = gg;
            a.B = nb; a.Ci = Cm; a.Co = Co; a.C1 = Cm; a.Co1 = Co; a.P = P; a.act_in = 1;
            a.act_cols = W; a.act_pitch = Wp; a.act_plane = (long long)Hp * Wp;
            a.vh_x = xg; a.vh_w = w1; a.vh_b = b1; a.vh_ci = Cin; a.vh_mode = 1;
            if (int rc = launch_channel_mix2(a, s)) return rc;
        }
        // 2. g_h = (w0^T gz) * gelu'(h)
        {
            ChannelMixArgs a{};
            a.x = gz; a.w = w0; a.y = gh; a.B = nb; a.Ci = Co; a.Co = Cm; a.C1 = Co; a.Co1 = Cm; a.P = P; a.transpose_w = 1;
            a.vh_x = xg; a.vh_w = w1; a.vh_b = b1; a.vh_ci = Cin; a.vh_mode = 2;
            if (int rc = launch_channel_mix2(a, s)) return rc;
        }
        // 3. fc0's weight / bias gradient: gz x gelu(h)^T;  4. fc_n1's: g_h x x^T
        if (int rc = launch_channel_wgrad_vh(gz, xg, w1, b1, Cin, gw0, gb0, wws, nb, Cm, Co, P, 1, s, acc)) return rc;
        if (int rc = launch_channel_wgrad2(gh, xg, nullptr, Cin, gw1, gb1, wws, nb, Cin, Cm, P, 0, acc, 0, s)) return rc;
    }
    return 0;
}

long long uno_channel_wgrad_ws_bytes(int B, int Ci, int Co, long long P) {
    if (B < 1 || Ci < 1 || Co < 1 || P < 1) return 0;
    return 4LL * channel_wgrad_ws_floats(B, Ci, Co, P, nullptr);
}

static int channel_wgrad_impl(const void* gy, const void* x, float* gw, float* gb, void* ws, int B, int Ci, int Co, long long P,
                              int act_x, int bf16, void* stream) {
    if (B < 0 || Ci < 1 || Co < 1 || P < 0) { set_error("uno_channel_wgrad: bad sizes B=%d Ci=%d Co=%d P=%lld", B, Ci, Co, P); return -1; }
    if (!gw) { set_error("uno_channel_wgrad: null pointer"); return -1; }
    if (B == 0 || P == 0) {
        if (hipMemsetAsync(gw, 0, sizeof(float) * Co * Ci, (hipStream_t)stream) != hipSuccess ||
            (gb && hipMemsetAsync(gb, 0, sizeof(float) * Co, (hipStream_t)stream) != hipSuccess)) { set_error("uno_channel_wgrad: memset failed"); return -5; }
        return 0;
    }
    if (!gy || !x || !ws) { set_error("uno_channel_wgrad: null pointer"); return -1; }
    return launch_channel_wgrad(gy, x, gw, gb, (float*)ws, B, Ci, Co, P, act_x, bf16, (hipStream_t)stream);
}

int uno_channel_wgrad(const float* gy, const float* x, float* gw, float* gb, void* ws, int B, int Ci, int Co, long long P,
                      int act_x, void* stream) {
    return channel_wgrad_impl(gy, x, gw, gb, ws, B, Ci, Co, P, act_x, 0, stream);
}

int uno_channel_wgrad_bf16(const void* gy, const void* x, float* gw, float* gb, void* ws, int B, int Ci, int Co, long long P,
                           int act_x, void* stream) {
    return channel_wgrad_impl(gy, x, gw, gb, ws, B, Ci, Co, P, act_x, 1, stream);
}

static int channel_wgrad2_impl(const void* gy, const void* x1, const void* x2, int C1, float* gw, float* gb, void* ws, int B, int Ci,
                               int Co, long long P, int act_x, int accumulate, int bf16, void* stream, const PixelWindow& win = PixelWindow()) {
    if (B < 0 || Ci < 1 || Co < 1 || P < 0) { set_error("uno_channel_wgrad2: bad sizes B=%d Ci=%d Co=%d P=%lld", B, Ci, Co, P); return -1; }
    if (accumulate < 0 || accumulate > 3 || accumulate == 2) { set_error("uno_channel_wgrad2: accumulate is 0, 1 or 3 (got %d)", accumulate); return -1; }
    if (!gw && accumulate != 3) { set_error("uno_channel_wgrad2: null pointer"); return -1; }
    if (B == 0 || P == 0) {
        if (accumulate == 3) {      // an empty call's partial sums are zeros
            if (!ws) { set_error("uno_channel_wgrad2: null pointer"); return -1; }
            if (hipMemsetAsync(ws, 0, uno_channel_wgrad_ws_bytes(B, Ci, Co, P), (hipStream_t)stream) != hipSuccess) { set_error("uno_channel_wgrad2: memset failed"); return -5; }
            return 0;
        }
        if (accumulate) return 0;
        if (hipMemsetAsync(gw, 0, sizeof(float) * Co * Ci, (hipStream_t)stream) != hipSuccess ||
            (gb && hipMemsetAsync(gb, 0, sizeof(float) * Co, (hipStream_t)stream) != hipSuccess)) { set_error("uno_channel_wgrad2: memset failed"); return -5; }
        return 0;
    }
    if (!gy || !x1 || !ws) { set_error("uno_channel_wgrad2: null pointer"); return -1; }
    return launch_channel_wgrad2(gy, x1, x2, x2 ? C1 : Ci, gw, gb, (float*)ws, B, Ci, Co, P, act_x, accumulate, bf16, (hipStream_t)stream, win);
}

int uno_channel_wgrad2_win(const float* gy, const float* x1, const float* x2, int C1, float* gw, float* gb, void* ws, int B, int Ci, int Co,
                           int rows, int cols, int pitch, long long plane, int act_x, int accumulate, void* stream) {
    PixelWindow win; long long P;
    if (!make_window("uno_channel_wgrad2_win", rows, cols, pitch, plane, &win, &P)) return -1;
    return channel_wgrad2_impl(gy, x1, x2, C1, gw, gb, ws, B, Ci, Co, P, act_x, accumulate, 0, stream, win);
}

int uno_channel_wgrad2(const float* gy, const float* x1, const float* x2, int C1, float* gw, float* gb, void* ws, int B, int Ci, int Co,
                       long long P, int act_x, int accumulate, void* stream) {
    return channel_wgrad2_impl(gy, x1, x2, C1, gw, gb, ws, B, Ci, Co, P, act_x, accumulate, 0, stream);
}

int uno_channel_wgrad2_bf16(const void* gy, const void* x1, const void* x2, int C1, float* gw, float* gb, void* ws, int B, int Ci, int Co,
                            long long P, int act_x, int accumulate, void* stream) {
    return channel_wgrad2_impl(gy, x1, x2, C1, gw, gb, ws, B, Ci, Co, P, act_x, accumulate, 1, stream);
}

int uno_channel_wgrad_finish(const void* parts, float* gw, float* gb, int Ci, int Co, long long nparts, int accumulate, void* stream) {
    if (Ci < 1 || Co < 1 || nparts < 1) { set_error("uno_channel_wgrad_finish: bad sizes Ci=%d Co=%d blocks=%lld", Ci, Co, nparts); return -1; }
    if (!parts || !gw) { set_error("uno_channel_wgrad_finish: null pointer"); return -1; }
    return launch_channel_wgrad_finish((const float*)parts, gw, gb, Ci, Co, nparts, accumulate, (hipStream_t)stream);
}

int uno_adam_step(float* p, const float* g, float* m, float* v, long long n, int is_complex, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int step, void* stream) {
    if (n < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) {
        set_error("uno_adam_step: bad arguments n=%lld step=%d betas=(%g, %g)", n, step, beta1, beta2);
        return -1;
    }
    if (n == 0) return 0;
    if (!p || !g || !m || !v) { set_error("uno_adam_step: null pointer"); return -1; }
    return launch_adam(p, g, m, v, n, is_complex, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

int uno_adam_step_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                        const long long* n, const int* is_complex, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int step, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!p || !g || !m || !v || !n || !is_complex))) {
        set_error("uno_adam_step_multi: bad arguments");
        return -1;
    }
    if (step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) {
        set_error("uno_adam_step_multi: bad arguments step=%d betas=(%g, %g)", step, beta1, beta2);
        return -1;
    }
    for (int t = 0; t < n_tensors; ++t) {
        if (n[t] < 0) { set_error("uno_adam_step_multi: tensor %d has n=%lld", t, n[t]); return -1; }
        if (n[t] > 0 && (!p[t] || !g[t] || !m[t] || !v[t])) { set_error("uno_adam_step_multi: null pointer (tensor %d)", t); return -1; }
    }
    // one launch per 24 tensors (csrc/adam.hip): the tensors' descriptors travel in the kernel arguments
    return launch_adam_multi(n_tensors, p, g, m, v, n, is_complex, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

int uno_adam_step_multi_dev(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                            const long long* n, const int* is_complex, double lr, double beta1, double beta2, double eps,
                            double weight_decay, int* step_counter, float* scalars, const double* hyper, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!p || !g || !m || !v || !n || !is_complex)) || !step_counter || !scalars) {
        set_error("uno_adam_step_multi_dev: bad arguments");
        return -1;
    }
    if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) { set_error("uno_adam_step_multi_dev: bad betas (%g, %g)", beta1, beta2); return -1; }
    for (int t = 0; t < n_tensors; ++t) {
        if (n[t] < 0) { set_error("uno_adam_step_multi_dev: tensor %d has n=%lld", t, n[t]); return -1; }
        if (n[t] > 0 && (!p[t] || !g[t] || !m[t] || !v[t])) { set_error("uno_adam_step_multi_dev: null pointer (tensor %d)", t); return -1; }
    }
    if (int rc = launch_adam_advance(step_counter, scalars, hyper, lr, eps, weight_decay, beta1, beta2, (hipStream_t)stream)) return rc;
    return launch_adam_multi(n_tensors, p, g, m, v, n, is_complex, lr, beta1, beta2, eps, weight_decay, 1, (hipStream_t)stream, scalars);
}

static int gelu_project_forward_impl(const void* pre, const float* w, const float* bias, void* out, int B, int C, long long P, int bf16, void* stream) {
    if (B < 0 || C < 1 || P < 0) { set_error("uno_gelu_project_forward: bad sizes B=%d C=%d P=%lld", B, C, P); return -1; }
    if (B == 0 || P == 0) return 0;
    if (!pre || !w || !out) { set_error("uno_gelu_project_forward: null pointer"); return -1; }
    return launch_gelu_project_fwd(pre, w, bias, out, B, C, P, bf16, (hipStream_t)stream);
}

int uno_gelu_project_forward(const float* pre, const float* w, const float* bias, float* out, int B, int C, long long P, void* stream) {
    return gelu_project_forward_impl(pre, w, bias, out, B, C, P, 0, stream);
}

int uno_gelu_project_forward_bf16(const void* pre, const float* w, const float* bias, void* out, int B, int C, long long P, void* stream) {
    return gelu_project_forward_impl(pre, w, bias, out, B, C, P, 1, stream);
}

long long uno_gelu_project_bwd_ws_bytes(int B, int C, long long P) {
    if (B < 1 || C < 1 || P < 1) return 0;
    return 4LL * gelu_project_ws_floats(B, C, P);
}

static int gelu_project_backward_impl(const void* pre, const float* w, const void* gout, void* gpre, float* gw, float* gb, void* ws, int B,
                                      int C, long long P, int bf16, void* stream, const PixelWindow& win = PixelWindow()) {
    if (B < 0 || C < 1 || P < 0) { set_error("uno_gelu_project_backward: bad sizes B=%d C=%d P=%lld", B, C, P); return -1; }
    if (!gw) { set_error("uno_gelu_project_backward: null pointer"); return -1; }
    if (B == 0 || P == 0) {
        if (hipMemsetAsync(gw, 0, sizeof(float) * C, (hipStream_t)stream) != hipSuccess ||
            (gb && hipMemsetAsync(gb, 0, sizeof(float), (hipStream_t)stream) != hipSuccess)) { set_error("uno_gelu_project_backward: memset failed"); return -5; }
        return 0;
    }
    if (!pre || !w || !gout || !gpre || !ws) { set_error("uno_gelu_project_backward: null pointer"); return -1; }
    return launch_gelu_project_bwd(pre, w, gout, gpre, gw, gb, (float*)ws, B, C, P, bf16, (hipStream_t)stream, win);
}

int uno_gelu_project_backward_win(const float* pre, const float* w, const float* gout, float* gpre, float* gw, float* gb, void* ws, int B,
                                  int C, int rows, int cols, int pitch, long long plane, void* stream) {
    PixelWindow win; long long P;
    if (!make_window("uno_gelu_project_backward_win", rows, cols, pitch, plane, &win, &P)) return -1;
    return gelu_project_backward_impl(pre, w, gout, gpre, gw, gb, ws, B, C, P, 0, stream, win);
}

int uno_gelu_project_backward(const float* pre, const float* w, const float* gout, float* gpre, float* gw, float* gb, void* ws, int B,
                              int C, long long P, void* stream) {
    return gelu_project_backward_impl(pre, w, gout, gpre, gw, gb, ws, B, C, P, 0, stream);
}

int uno_gelu_project_backward_bf16(const void* pre, const float* w, const void* gout, void* gpre, float* gw, float* gb, void* ws, int B,
                                   int C, long long P, void* stream) {
    return gelu_project_backward_impl(pre, w, gout, gpre, gw, gb, ws, B, C, P, 1, stream);
}

static int gelu_pad_impl(const void* s, const void* gy, void* out, int n_img, int H, int W, int Hp, int Wp, int backward, int bf16, void* stream) {
    if (n_img < 0 || H < 1 || W < 1 || Hp < H || Wp < W) { set_error("uno_gelu_pad: bad sizes (%d, %d) -> (%d, %d)", H, W, Hp, Wp); return -1; }
    if (n_img == 0) return 0;
    if (!s || !out || (backward && !gy)) { set_error("uno_gelu_pad: null pointer"); return -1; }
    return launch_gelu_pad(s, gy, out, n_img, H, W, Hp, Wp, backward, bf16, (hipStream_t)stream);
}

int uno_gelu_pad(const float* s, const float* gy, float* out, int n_img, int H, int W, int Hp, int Wp, int backward, void* stream) {
    return gelu_pad_impl(s, gy, out, n_img, H, W, Hp, Wp, backward, 0, stream);
}

int uno_gelu_pad_bf16(const void* s, const void* gy, void* out, int n_img, int H, int W, int Hp, int Wp, int backward, void* stream) {
    return gelu_pad_impl(s, gy, out, n_img, H, W, Hp, Wp, backward, 1, stream);
}

int uno_transpose_batched(const float* in, float* out, int B, long long R, int C, long long ld_in, long long sb_in, long long ld_out,
                          long long sb_out, void* stream) {
    if (B < 0 || R < 0 || C < 0 || ld_in < C || ld_out < R || sb_in < 0 || sb_out < 0) {
        set_error("uno_transpose_batched: bad sizes (B %d, R %lld, C %d, pitches %lld / %lld)", B, R, C, ld_in, ld_out);
        return -1;
    }
    if (B == 0 || R == 0 || C == 0) return 0;
    if (!in || !out) { set_error("uno_transpose_batched: null pointer"); return -1; }
    return launch_transpose_batched(in, out, B, R, C, ld_in, sb_in, ld_out, sb_out, (hipStream_t)stream);
}

static int instnorm_forward_impl(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long long rows, int C,
                                 long long N, float eps, int gelu, int bf16, void* stream) {
    if (rows < 0 || C < 1 || N < 1 || (rows % C) != 0) { set_error("uno_instnorm_forward: bad sizes rows=%lld C=%d N=%lld", rows, C, N); return -1; }
    if (rows == 0) return 0;
    if (!x || !y || !mean || !rstd) { set_error("uno_instnorm_forward: null pointer"); return -1; }
    return launch_instnorm_fwd(x, gamma, beta, y, mean, rstd, rows, C, N, eps, gelu, bf16, (hipStream_t)stream);
}

int uno_instnorm_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long long rows, int C,
                         long long N, float eps, int gelu, void* stream) {
    return instnorm_forward_impl(x, gamma, beta, y, mean, rstd, rows, C, N, eps, gelu, 0, stream);
}

int uno_instnorm_forward_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long long rows, int C,
                              long long N, float eps, int gelu, void* stream) {
    return instnorm_forward_impl(x, gamma, beta, y, mean, rstd, rows, C, N, eps, gelu, 1, stream);
}

static int instnorm_backward_impl(const void* x, const void* gy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                  void* gx, float* s1, float* s2, long long rows, int C, long long N, int gelu, int bf16, void* stream) {
    if (rows < 0 || C < 1 || N < 1 || (rows % C) != 0) { set_error("uno_instnorm_backward: bad sizes rows=%lld C=%d N=%lld", rows, C, N); return -1; }
    if (rows == 0) return 0;
    if (!x || !gy || !mean || !rstd || !gx || !s1 || !s2) { set_error("uno_instnorm_backward: null pointer"); return -1; }
    return launch_instnorm_bwd(x, gy, gamma, beta, mean, rstd, gx, s1, s2, rows, C, N, gelu, bf16, (hipStream_t)stream);
}

int uno_instnorm_backward(const float* x, const float* gy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                          float* gx, float* s1, float* s2, long long rows, int C, long long N, int gelu, void* stream) {
    return instnorm_backward_impl(x, gy, gamma, beta, mean, rstd, gx, s1, s2, rows, C, N, gelu, 0, stream);
}

int uno_instnorm_backward_bf16(const void* x, const void* gy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                               void* gx, float* s1, float* s2, long long rows, int C, long long N, int gelu, void* stream) {
    return instnorm_backward_impl(x, gy, gamma, beta, mean, rstd, gx, s1, s2, rows, C, N, gelu, 1, stream);
}

int uno_cdft_axis(const float* in, float* out, int inverse, int n_img, int H, int m1, int m2, int m3, float scale,
                  int mask_overlap, void* stream) {
    if (n_img < 0 || H < 1 || m1 < 1 || m1 > H || m2 < 1 || m3 < 1) {
        set_error("uno_cdft_axis: bad sizes n_img=%d H=%d modes=(%d,%d,%d)", n_img, H, m1, m2, m3);
        return -1;
    }
    if (n_img == 0) return 0;
    if (!in || !out) { set_error("uno_cdft_axis: null pointer"); return -1; }
    CdftParams p;
    p.in = in; p.out = out; p.n_img = n_img; p.H = H; p.C = 2 * m2 * m3; p.m1 = m1; p.m2 = m2; p.m3 = m3;
    p.scale = scale; p.mask = mask_overlap ? 1 : 0; p.rowfreq = nullptr;
    p.tw = twiddle_table(H);
    if (!p.tw) return -6;
    if (m1 > 40) return launch_cdft_generic(p, inverse != 0, (hipStream_t)stream);
    return launch_cdft(p, inverse != 0, (hipStream_t)stream);
}

// ---- FFT crop / resample of pointwise_op_3D (reference integral_operators.py:448-463) as pruned transforms with explicit
// frequency tables.  Along a complex axis of length N resampled to M the reference keeps the spectrum INDICES
// r in ([0, M/2) u [N - M/2, N)) n [0, min(N, M)) and irfftn reads index r as frequency r of a length-M transform (its trimming /
// zero-padding happens at the end of the axis): forward frequency f_in[j] = r_j on N points, inverse frequency f_out[j] = r_j on
// M points - the binding builds the tables, so this entry point is the general "pruned DFT - pruned inverse DFT" pair.
long long uno_fft_resample3d_ws_bytes(int n_vol, int D1, int M1, int J1, int J2, int m3) {
    const long long C = (long long)J2 * m3;
    return 8LL * n_vol * ((long long)D1 * C + (long long)M1 * C + (long long)J1 * C);
}

static int fft_resample3d_impl(const float* x, float* y, void* ws, int n_vol, int D1, int D2, int D3, int M1, int M2, int M3,
                               int J1, const int* f1_in, const int* f1_out, int J2, const int* f2_in, const int* f2_out, int m3,
                               float scale, int herm_in, int herm_out, int accumulate, float* act_out, void* stream) {
    const char* who = "uno_fft_resample3d";
    if (n_vol < 0 || D1 < 1 || D2 < 1 || D3 < 1 || M1 < 1 || M2 < 1 || M3 < 1) { set_error("%s: bad sizes", who); return -1; }
    if (J1 < 2 || (J1 & 1) || J2 < 2 || (J2 & 1) || J1 > 80 || J2 > 48 || m3 < 1 || m3 > D3 / 2 + 1 || m3 > M3 / 2 + 1) {
        set_error("%s: row counts must be even (J1=%d <= 80, J2=%d <= 48) and 1 <= modes3=%d <= n/2+1", who, J1, J2, m3);
        return -1;
    }
    if (n_vol == 0) return 0;
    if (!x || !y || !ws || !f1_in || !f1_out || !f2_in || !f2_out) { set_error("%s: null pointer", who); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const long long C = (long long)J2 * m3;
    float* Z1 = static_cast<float*>(ws);                          // (n_vol * D1, J2, m3) c64
    float* Z2 = Z1 + 2LL * n_vol * D1 * C;                        // (n_vol * M1, J2, m3) c64
    float* S = Z2 + 2LL * n_vol * M1 * C;                         // (n_vol, 4, J1/2, J2/2, m3) c64
    Dft2dParams p;
    p.n_img = n_vol * D1; p.H = D2; p.W = D3; p.m1 = J2 / 2; p.m2 = m3; p.scale = 1.0f; p.herm = herm_in ? 1 : 0; p.mask = 0; p.bf16 = 0; p.nw = 1; p.exp = 0; p.accumulate = 0; p.act_out = nullptr;
    p.sp_group = p.n_img; p.sp_stride = 0; p.sp_offset = 0;
    p.in = x; p.out = Z1; p.rowfreq = f2_in;
    p.twH = twiddle_table(D2); p.twW = twiddle_table(D3);
    if (!p.twH || !p.twW) return -6;
    if (!dft2d_fwd_plane_applies(p)) { set_error("%s: input planes %d x %d (%d of them) are outside the plane-batched kernels' range", who, D2, D3, p.n_img); return -2; }
    if (int rc = launch_dft2d_fwd_plane(p, s)) return rc;
    CdftParams c;
    c.n_img = n_vol; c.C = (int)C; c.m1 = J1 / 2; c.m2 = J2 / 2; c.m3 = m3; c.mask = 0; c.scale = 1.0f;
    c.in = Z1; c.out = S; c.H = D1; c.rowfreq = f1_in; c.tw = twiddle_table(D1);
    if (!c.tw) return -6;
    if (int rc = launch_cdft(c, false, s)) return rc;
    c.in = S; c.out = Z2; c.H = M1; c.rowfreq = f1_out; c.tw = twiddle_table(M1);
    if (!c.tw) return -6;
    if (int rc = launch_cdft(c, true, s)) return rc;
    p.n_img = n_vol * M1; p.H = M2; p.W = M3; p.scale = scale; p.herm = herm_out ? 1 : 0;
    p.sp_group = p.n_img;
    p.in = Z2; p.out = y; p.rowfreq = f2_out;
    p.twH = twiddle_table(M2); p.twW = twiddle_table(M3);
    if (!p.twH || !p.twW) return -6;
    if (!dft2d_inv_plane_applies(p)) { set_error("%s: output planes %d x %d (%d of them) are outside the plane-batched kernels' range", who, M2, M3, p.n_img); return -2; }
    p.accumulate = accumulate ? 1 : 0; p.act_out = act_out;
    return launch_dft2d_inv_plane(p, s);
}

int uno_fft_resample3d(const float* x, float* y, void* ws, int n_vol, int D1, int D2, int D3, int M1, int M2, int M3,
                       int J1, const int* f1_in, const int* f1_out, int J2, const int* f2_in, const int* f2_out, int m3,
                       float scale, int herm_in, int herm_out, void* stream) {
    return fft_resample3d_impl(x, y, ws, n_vol, D1, D2, D3, M1, M2, M3, J1, f1_in, f1_out, J2, f2_in, f2_out, m3, scale, herm_in, herm_out,
                               0, nullptr, stream);
}

int uno_fft_resample3d_acc(const float* x, float* y, float* y_act, void* ws, int n_vol, int D1, int D2, int D3, int M1, int M2, int M3,
                           int J1, const int* f1_in, const int* f1_out, int J2, const int* f2_in, const int* f2_out, int m3,
                           float scale, int herm_in, int herm_out, void* stream) {
    return fft_resample3d_impl(x, y, ws, n_vol, D1, D2, D3, M1, M2, M3, J1, f1_in, f1_out, J2, f2_in, f2_out, m3, scale, herm_in, herm_out,
                               1, y_act, stream);
}

static int check_modes3d(const char* who, int H, int W, int T, int Ho, int Wo, int To, int m1, int m2, int m3) {
    if (H < 1 || W < 1 || T < 1 || Ho < 1 || Wo < 1 || To < 1) { set_error("%s: empty grid", who); return -1; }
    if (m1 < 1 || m1 > H || m1 > Ho) { set_error("%s: modes1=%d incompatible with axis %d -> %d", who, m1, H, Ho); return -1; }
    if (m2 < 1 || m2 > W || m2 > Wo) { set_error("%s: modes2=%d incompatible with axis %d -> %d", who, m2, W, Wo); return -1; }
    if (m3 < 1 || m3 > T / 2 + 1 || m3 > To / 2 + 1) {
        set_error("%s: modes3=%d incompatible with axis %d -> %d (need modes3 <= n/2+1)", who, m3, T, To);
        return -1;
    }
    return 0;
}

// volumes (n_vol, D1, D2, D3) -> corner-major truncated spectra (n_vol, 4, m1, m2, m3); `adjoint` = the Hermitian-weighted, masked form
// the backward pass applies to the output gradient.  One workgroup per volume where that fits (K1v), else plane by plane (K1p) into
// the workspace Z (n_vol * D1, 2 m2, m3) c64 and the leading axis from there (K5).
static int fwd_transform3d(const float* x, float* spec, float* Z, int n_vol, int D1, int D2, int D3, int m1, int m2, int m3, float scale,
                           int adjoint, hipStream_t s) {
    if (vol3d_fwd_applies(n_vol, D1, D2, D3, m1, m2, m3)) {
        Vol3dParams v;
        v.in = x; v.out = spec; v.n_vol = n_vol; v.D1 = D1; v.D2 = D2; v.D3 = D3; v.m1 = m1; v.m2 = m2; v.m3 = m3;
        v.scale = scale; v.herm = adjoint;
        v.tw1 = twiddle_table(2 * D1); v.tw2 = twiddle_table(2 * D2); v.tw3 = twiddle_table(D3);
        if (!v.tw1 || !v.tw2 || !v.tw3) return -6;
        return launch_dft3d_fwd_volume(v, s);
    }
    if (int rc = dft2d(false, x, Z, n_vol * D1, D2, D3, m2, m3, scale, adjoint, adjoint, s)) return rc;
    return uno_cdft_axis(Z, spec, 0, n_vol, D1, m1, m2, m3, 1.0f, adjoint, (void*)s);
}

// the inverse: corner-major spectra -> volumes; `weighted` = Hermitian weights + later-wins masks (the forward pass's irfftn)
static int inv_transform3d(const float* spec, float* y, float* Z, int n_vol, int D1, int D2, int D3, int m1, int m2, int m3, float scale,
                           int weighted, hipStream_t s) {
    if (vol3d_inv_applies(n_vol, D1, D2, D3, m1, m2, m3)) {
        Vol3dParams v;
        v.in = spec; v.out = y; v.n_vol = n_vol; v.D1 = D1; v.D2 = D2; v.D3 = D3; v.m1 = m1; v.m2 = m2; v.m3 = m3;
        v.scale = scale; v.herm = weighted;
        v.tw1 = twiddle_table(2 * D1); v.tw2 = twiddle_table(2 * D2); v.tw3 = twiddle_table(D3);
        if (!v.tw1 || !v.tw2 || !v.tw3) return -6;
        return launch_dft3d_inv_volume(v, s);
    }
    if (int rc = uno_cdft_axis(spec, Z, 1, n_vol, D1, m1, m2, m3, 1.0f, weighted, (void*)s)) return rc;
    return dft2d(true, Z, y, n_vol * D1, D2, D3, m2, m3, scale, weighted, weighted, s);
}

long long uno_spectral_conv3d_fwd_ws_bytes(int B, int Ci, int Co, int H, int Ho, int m1, int m2, int m3) {
    const long long C = 2LL * m2 * m3;
    return 8LL * B * ((long long)Ci * H * C + (long long)Co * Ho * C + 4LL * Co * m1 * m2 * m3);
}

long long uno_spectral_conv3d_bwd_ws_bytes(int B, int Ci, int Co, int H, int Ho, int m1, int m2, int m3) {
    const long long C = 2LL * m2 * m3;
    return 8LL * B * ((long long)Ci * H * C + (long long)Co * Ho * C + 4LL * (Ci + Co) * m1 * m2 * m3);
}

int uno_spectral_conv3d_forward(const float* x, const float* const* w, float* y, float* xtrunc, void* ws, int B, int Ci,
                                int Co, int H, int W, int T, int Ho, int Wo, int To, int m1, int m2, int m3, void* stream) {
    const char* who = "uno_spectral_conv3d_forward";
    if (B < 0 || Ci < 1 || Co < 1) { set_error("%s: bad sizes B=%d Ci=%d Co=%d", who, B, Ci, Co); return -1; }
    if (int rc = check_modes3d(who, H, W, T, Ho, Wo, To, m1, m2, m3)) return rc;
    if (B == 0) return 0;
    if (!x || !w || !y || !xtrunc || !ws) { set_error("%s: null pointer", who); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const long long C = 2LL * m2 * m3, Mc = (long long)m1 * m2 * m3;
    float* Z1 = static_cast<float*>(ws);                         // (B*Ci*H, 2 m2, m3) c64
    float* Z2 = Z1 + 2LL * B * Ci * H * C;                        // (B*Co*Ho, 2 m2, m3) c64
    float* O5 = Z2 + 2LL * B * Co * Ho * C;                       // (B, Co, 4, m1, m2, m3) c64
    const float inv_n = 1.0f / ((float)H * (float)W * (float)T);
    // rfftn over (W, T) plane by plane, then the H axis                       (reference :398)
    if (int rc = fwd_transform3d(x, xtrunc, Z1, B * Ci, H, W, T, m1, m2, m3, inv_n, 0, s)) return rc;
    // four corner einsums "bixyz,ioxyz->boxyz"                                  (reference :410-421)
    if (int rc = uno_mode_mix(xtrunc, w, O5, 0, B, Ci, Co, 4, (int)Mc, stream)) return rc;
    // irfftn(out_ft, s=(Ho, Wo, To), norm="forward"); later-wins masks are separable per axis (reference :400-426)
    return inv_transform3d(O5, y, Z2, B * Co, Ho, Wo, To, m1, m2, m3, 1.0f, 1, s);
}

int uno_spectral_conv3d_backward(const float* gy, const float* xtrunc, const float* const* w, float* gx, float* const* gw,
                                 void* ws, int B, int Ci, int Co, int H, int W, int T, int Ho, int Wo, int To, int m1,
                                 int m2, int m3, void* stream) {
    const char* who = "uno_spectral_conv3d_backward";
    if (B < 0 || Ci < 1 || Co < 1) { set_error("%s: bad sizes B=%d Ci=%d Co=%d", who, B, Ci, Co); return -1; }
    if (int rc = check_modes3d(who, H, W, T, Ho, Wo, To, m1, m2, m3)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const long long C = 2LL * m2 * m3, Mc = (long long)m1 * m2 * m3;
    if (B == 0) {
        if (gw)
            for (int c = 0; c < 4; ++c)
                if (hipMemsetAsync(gw[c], 0, 8ULL * Ci * Co * Mc, s) != hipSuccess) { set_error("memset failed"); return -5; }
        return 0;
    }
    if (!gy || !xtrunc || !w || !ws) { set_error("%s: null pointer", who); return -1; }
    float* Z1 = static_cast<float*>(ws);
    float* Z2 = Z1 + 2LL * B * Ci * H * C;
    float* gO = Z2 + 2LL * B * Co * Ho * C;
    float* gX = gO + 2LL * B * Co * 4 * Mc;
    if (int rc = fwd_transform3d(gy, gO, Z2, B * Co, Ho, Wo, To, m1, m2, m3, 1.0f, 1, s)) return rc;
    // The weight gradient stays on the caller's stream (measured round 3, one box, A/B: on the side stream next to the
    // input-gradient GEMM and the inverse transform - the 2-D arrangement - the C4 block backward took 127 us against 119.5 us
    // in sequence: at 4 corners of weights the two per-mode GEMMs are each bound by the same weight / spectrum streams)
    // (round 6, measured and not adopted here: both GEMMs in one launch - uno_mode_backward, what the 2-D layers use - took 48.0 us at the
    // C4 block against 24.5 + 21.0 in sequence: with four corners of weights each role fills the chip on its own)
    if (gw)
        if (int rc = uno_mode_wgrad(xtrunc, gO, gw, B, Ci, Co, 4, (int)Mc, stream)) return rc;
    if (gx) {
        if (int rc = uno_mode_mix(gO, w, gX, 1, B, Ci, Co, 4, (int)Mc, stream)) return rc;
        const float inv_n = 1.0f / ((float)H * (float)W * (float)T);
        if (int rc = inv_transform3d(gX, gx, Z1, B * Ci, H, W, T, m1, m2, m3, inv_n, 0, s)) return rc;
    }
    return 0;
}

static int spectral_conv2d_forward(const float* x, const float* w1, const float* w2, float* y, float* xtrunc, void* ws,
                                   int B, int Ci, int Co, int H, int W, int Ho, int Wo, int m1, int m2, void* stream, int bf16, int w_half = 0) {
    if (B < 0 || Ci < 1 || Co < 1) { set_error("uno_spectral_conv2d_forward: bad sizes B=%d Ci=%d Co=%d", B, Ci, Co); return -1; }
    if (int rc = check_modes2d("uno_spectral_conv2d_forward", H, W, Ho, Wo, m1, m2)) return rc;
    if (B == 0) return 0;           // empty batch: nothing to do (empty tensors carry null pointers)
    if (!x || !w1 || !w2 || !y || !xtrunc || !ws) { set_error("uno_spectral_conv2d_forward: null pointer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    float* O = static_cast<float*>(ws);
    // rfft2(x, norm="forward") restricted to the two corners            (reference :187)
    if (int rc = dft2d(false, x, xtrunc, B * Ci, H, W, m1, m2, 1.0f / ((float)H * (float)W), 0, 0, s, 0, 0, 0, bf16)) return rc;
    // einsum("bixy,ioxy->boxy") with weights1 / weights2                  (reference :198-203)
    const float* wv[2] = {w1, w2};
    if (int rc = mode_mix_impl(xtrunc, wv, O, 0, B, Ci, Co, 2, m1 * m2, stream, w_half)) return rc;
    // irfft2(out_ft, s=(Ho, Wo), norm="forward"), later-wins on overlapping rows (reference :190-206)
    return dft2d(true, O, y, B * Co, Ho, Wo, m1, m2, 1.0f, 1, 1, s, 0, 0, 0, bf16);
}

int uno_spectral_conv2d_forward(const float* x, const float* w1, const float* w2, float* y, float* xtrunc, void* ws,
                                int B, int Ci, int Co, int H, int W, int Ho, int Wo, int m1, int m2, void* stream) {
    return spectral_conv2d_forward(x, w1, w2, y, xtrunc, ws, B, Ci, Co, H, W, Ho, Wo, m1, m2, stream, 0);
}

int uno_spectral_conv2d_forward_bf16(const void* x, const float* w1, const float* w2, void* y, float* xtrunc, void* ws,
                                     int B, int Ci, int Co, int H, int W, int Ho, int Wo, int m1, int m2, void* stream) {
    return spectral_conv2d_forward(static_cast<const float*>(x), w1, w2, static_cast<float*>(y), xtrunc, ws, B, Ci, Co, H, W, Ho, Wo,
                                   m1, m2, stream, 1);
}

static int spectral_conv2d_backward(const float* gy, const float* xtrunc, const float* w1, const float* w2, float* gx,
                                    float* gw1, float* gw2, void* ws, int B, int Ci, int Co, int H, int W, int Ho, int Wo,
                                    int m1, int m2, void* stream, int bf16, int w_half = 0, int accumulate_gw = 0) {
    if (B > 0 && (!gy || !xtrunc || !w1 || !w2 || !ws)) { set_error("uno_spectral_conv2d_backward: null pointer"); return -1; }
    if ((gw1 == nullptr) != (gw2 == nullptr)) { set_error("uno_spectral_conv2d_backward: gw1/gw2 must both be given or both be NULL"); return -1; }
    if (B < 0 || Ci < 1 || Co < 1) { set_error("uno_spectral_conv2d_backward: bad sizes B=%d Ci=%d Co=%d", B, Ci, Co); return -1; }
    if (int rc = check_modes2d("uno_spectral_conv2d_backward", H, W, Ho, Wo, m1, m2)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const long long P = 2LL * m1 * m2;
    float* gO = static_cast<float*>(ws);
    float* gX = gO + 2LL * B * Co * P;
    if (B == 0) {
        if (gw1 && !accumulate_gw) {
            if (hipMemsetAsync(gw1, 0, 8ULL * Ci * Co * m1 * m2, s) != hipSuccess || hipMemsetAsync(gw2, 0, 8ULL * Ci * Co * m1 * m2, s) != hipSuccess) {
                set_error("memset failed"); return -5;
            }
        }
        return 0;
    }
    // gO = c (.) keep (.) DFT_trunc(gy)                                   (adjoint of irfft2 + CopySlices)
    if (int rc = dft2d(false, gy, gO, B * Co, Ho, Wo, m1, m2, 1.0f, 1, 1, s, 0, 0, 0, bf16)) return rc;
    // The weight gradient only shares gO with the input-gradient chain: it runs on the side stream next to the
    // (under-filled) input-gradient GEMM and the store-bound inverse DFT.
    // (round 6, measured and not adopted HERE: both per-mode GEMMs in one launch - uno_mode_backward, what the stage-by-stage callers use -
    // took the C2 block backward from 352-355 to 361 us: this composite already hides the weight gradient behind the inverse transform)
    SideStream* side = (gw1 && gx) ? side_stream_of_current_device() : nullptr;
    int rc_w = 0;
    if (gw1) {
        float* gwv[2] = {gw1, gw2};
        if (side) {
            side->mu.lock();
            if (hipEventRecord(side->fork_ev, s) != hipSuccess || hipStreamWaitEvent(side->s, side->fork_ev, 0) != hipSuccess) {
                side->mu.unlock();
                side = nullptr;
            }
        }
        rc_w = mode_wgrad_impl(xtrunc, gO, gwv, B, Ci, Co, 2, m1 * m2, accumulate_gw, side ? (void*)side->s : stream);
    }
    int rc_x = 0;
    if (gx && rc_w == 0) {
        const float* wv[2] = {w1, w2};
        rc_x = mode_mix_impl(gO, wv, gX, 1, B, Ci, Co, 2, m1 * m2, stream, w_half);
        // gx = 1/(H W) Re iDFT_trunc(gX)                                   (adjoint of rfft2(norm="forward"))
        if (rc_x == 0) rc_x = dft2d(true, gX, gx, B * Ci, H, W, m1, m2, 1.0f / ((float)H * (float)W), 0, 0, s, 0, 0, 0, bf16);
    }
    if (side) {
        const bool joined = hipEventRecord(side->join_ev, side->s) == hipSuccess && hipStreamWaitEvent(s, side->join_ev, 0) == hipSuccess;
        side->mu.unlock();
        if (!joined) { set_error("uno_spectral_conv2d_backward: side-stream join failed"); return -5; }
    }
    return rc_w ? rc_w : rc_x;
}

int uno_spectral_conv2d_backward(const float* gy, const float* xtrunc, const float* w1, const float* w2, float* gx,
                                 float* gw1, float* gw2, void* ws, int B, int Ci, int Co, int H, int W, int Ho, int Wo,
                                 int m1, int m2, void* stream) {
    return spectral_conv2d_backward(gy, xtrunc, w1, w2, gx, gw1, gw2, ws, B, Ci, Co, H, W, Ho, Wo, m1, m2, stream, 0);
}

int uno_spectral_conv2d_backward_bf16(const void* gy, const float* xtrunc, const float* w1, const float* w2, void* gx,
                                      float* gw1, float* gw2, void* ws, int B, int Ci, int Co, int H, int W, int Ho, int Wo,
                                      int m1, int m2, void* stream) {
    return spectral_conv2d_backward(static_cast<const float*>(gy), xtrunc, w1, w2, static_cast<float*>(gx), gw1, gw2, ws, B, Ci, Co,
                                    H, W, Ho, Wo, m1, m2, stream, 1);
}

int uno_spectral_conv2d_backward_acc(const void* gy, const float* xtrunc, const void* w1, const void* w2, void* gx,
                                     float* gw1, float* gw2, void* ws, int B, int Ci, int Co, int H, int W, int Ho, int Wo,
                                     int m1, int m2, int io_format, int accumulate_gw, void* stream) {
    if (io_format < 0 || io_format > 2) { set_error("uno_spectral_conv2d_backward_acc: io_format %d (0 f32, 1 bf16, 2 bf16 + fp16 weights)", io_format); return -1; }
    return spectral_conv2d_backward(static_cast<const float*>(gy), xtrunc, static_cast<const float*>(w1), static_cast<const float*>(w2),
                                    static_cast<float*>(gx), gw1, gw2, ws, B, Ci, Co, H, W, Ho, Wo, m1, m2, stream, io_format >= 1,
                                    io_format == 2, accumulate_gw ? 1 : 0);
}

int uno_spectral_conv2d_forward_mixed(const void* x, const void* w1, const void* w2, void* y, float* xtrunc, void* ws,
                                      int B, int Ci, int Co, int H, int W, int Ho, int Wo, int m1, int m2, void* stream) {
    return spectral_conv2d_forward(static_cast<const float*>(x), static_cast<const float*>(w1), static_cast<const float*>(w2),
                                   static_cast<float*>(y), xtrunc, ws, B, Ci, Co, H, W, Ho, Wo, m1, m2, stream, 1, 1);
}

int uno_spectral_conv2d_backward_mixed(const void* gy, const float* xtrunc, const void* w1, const void* w2, void* gx,
                                       float* gw1, float* gw2, void* ws, int B, int Ci, int Co, int H, int W, int Ho, int Wo,
                                       int m1, int m2, void* stream) {
    return spectral_conv2d_backward(static_cast<const float*>(gy), xtrunc, static_cast<const float*>(w1), static_cast<const float*>(w2),
                                    static_cast<float*>(gx), gw1, gw2, ws, B, Ci, Co, H, W, Ho, Wo, m1, m2, stream, 1, 1);
}

}  // extern "C"
