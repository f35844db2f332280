// Shared device/host helpers for the U-NO spectral-convolution kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uno {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16), exact f32 (k-ordered fmaf chain).
// Lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// lane l, register r of C/D holds D[row = 4 * (l >> 4) + r][col = l & 15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// erf in float32 as ONE branch-free rational function (odd degree-13 / even degree-8 in the clamped argument, the approximant of
// Eigen's / XLA's float erf): 12 fmas, a reciprocal with one Newton step, no exp.  Maximum error 4.5e-7 (absolute and relative)
// over the real line, i.e. the exact-erf GELU 0.5 x (1 + erf(x / sqrt 2)) of the reference (F.gelu, integral_operators.py:282)
// comes out with the SAME error against float64 as torch's own float32 kernel (relative L2 7e-8, largest element error 1.4e-6:
// measured on 2 M points, tools/dev/erf_accuracy.py).  The library erff is a branching piecewise form (|x| < 1 polynomial, else
// exp-based): the lanes of a wave take both paths, ~3x the instructions - and the GELU forms sit on the load / store paths of
// K8, K9, K11, K12, K13, where they were the co-limiter (the fused-GELU instantiations ran 20-50 % behind the plain ones).
__device__ __forceinline__ float uno_erf(float x) {
    x = __builtin_fminf(__builtin_fmaxf(x, -4.f), 4.f);          // |x| >= 4: erf = +-1 in float32
    const float t = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, t, 2.77068142495902e-08f);
    p = fmaf(p, t, -2.10102402082508e-06f);
    p = fmaf(p, t, -5.69250639462346e-05f);
    p = fmaf(p, t, -7.34990630326855e-04f);
    p = fmaf(p, t, -2.95459980854025e-03f);
    p = fmaf(p, t, -1.60960333262415e-02f);
    p *= x;
    float q = -1.45660718464996e-05f;
    q = fmaf(q, t, -2.13374055278905e-04f);
    q = fmaf(q, t, -1.68282697438203e-03f);
    q = fmaf(q, t, -7.37332916720468e-03f);
    q = fmaf(q, t, -1.42647390514189e-02f);
    float r = __builtin_amdgcn_rcpf(q);
    r = r * fmaf(-q, r, 2.f);                                    // one Newton step: the quotient is correctly rounded to ~0.5 ulp
    return p * r;
}
// exact-erf GELU and its derivative (Phi(x) + x phi(x)); exp through v_exp_f32 (2 ulp: the term is <= 0.4 |x| e^{-x^2 / 2})
__device__ __forceinline__ float uno_gelu(float x) { return 0.5f * x * (1.f + uno_erf(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float uno_dgelu(float x) {
    const float cdf = 0.5f * (1.f + uno_erf(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return fmaf(x, pdf, cdf);
}

// ---- pixel windows (round 5).  The channel-mix / weight-gradient / GELU-projection calls see a (B, C, P) tensor whose pixel axis is
// dense.  With a WINDOW the P = rows x cols logical pixels of a call are the top-left corner of a wider plane: logical pixel q lives
// at element (q / cols) * pitch + q % cols of its channel plane and channel planes are `plane` elements apart (the 421 x 421 domain
// inside the 446 x 446 padded grid of the Darcy model: fc1, its GELU and fc2 work on the domain only, reference
// darcy_flow_uno2d.py:126-131 crops first).  cols is a multiple of 4 - a lane's four pixels never straddle a row; the columns
// between the domain and the next multiple of 4 are processed like domain pixels (they exist: pitch >= cols) - and 260 <= cols,
// rows * cols < 2^24: q / cols = (q * ceil(2^40 / cols)) >> 40 exactly, and any run of up to 256 pixels touches two rows at most.
struct PixelWindow { long long plane = 0; int cols = 0, pitch = 0; };       // cols == 0: dense
struct PixMap {             // kernel-side form: plane stride + the window (rl == 0: dense, PS == P)
    int PS; int rl, skip; unsigned magic;
};
inline PixMap pix_map(const PixelWindow& w, long long P) {
    if (!w.cols) return PixMap{(int)P, 0, 0, 0u};
    return PixMap{(int)w.plane, w.cols, w.pitch - w.cols, (unsigned)(((1ULL << 40) + w.cols - 1) / (unsigned long long)w.cols)};
}
// nullptr, or why the window cannot be used with P logical pixels
inline const char* pix_window_error(const PixelWindow& w, long long P) {
    if (!w.cols) return nullptr;
    if (w.cols % 4 || w.cols < 260 || w.pitch < w.cols) return "window: cols must be a multiple of 4, >= 260 and <= pitch";
    if (P % w.cols || P >= (1LL << 24)) return "window: the pixel count must be rows * cols, below 2^24";
    if (w.plane < (P / w.cols - 1) * (long long)w.pitch + w.cols || w.plane > 0x7fffffffLL) return "window: the plane does not hold rows * pitch elements";
    return nullptr;
}
// physical offsets of a run of <= 256 logical pixels starting at the (wave-uniform) pixel q0
struct PixRun {
    int base, bound, skip;
    __device__ __forceinline__ int operator()(int q) const { return q + base + (q >= bound ? skip : 0); }
};
__device__ __forceinline__ PixRun pix_run(const PixMap& m, int q0) {
    if (m.rl == 0) return PixRun{0, 0x7fffffff, 0};
    const int r = (int)(((unsigned long long)(unsigned)q0 * m.magic) >> 40);
    return PixRun{r * m.skip, (r + 1) * m.rl, m.skip};
}
__device__ __forceinline__ int pix_phys(const PixMap& m, int q) {           // any lane-varying pixel
    if (m.rl == 0) return q;
    return q + (int)(((unsigned long long)(unsigned)q * m.magic) >> 40) * m.skip;
}

// idx, inc and lim are byte offsets into a float2 table of lim/8 entries; idx < lim, inc < lim.
__device__ __forceinline__ unsigned wrap_add(unsigned idx, unsigned inc, unsigned lim) {
    unsigned t = idx + inc;
    return min(t, t - lim);          // t - lim wraps to a huge value when t < lim
}
__device__ __forceinline__ unsigned wrap_sub(unsigned idx, unsigned dec, unsigned lim) {
    unsigned t = idx - dec;          // wraps when idx < dec
    return min(t, t + lim);
}

// 4-byte-aligned 16-byte load: rows of odd length leave row starts only 4-byte aligned.
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

// bfloat16 activations (config C5): four consecutive elements, 2-byte aligned; widened by << 16, narrowed round-to-nearest-even
struct __attribute__((packed, aligned(2))) h4u { unsigned short v[4]; };
template <bool BF16> struct IoElem { typedef float type; typedef f4u vec4; };
template <> struct IoElem<true> { typedef unsigned short type; typedef h4u vec4; };
__device__ __forceinline__ float io_widen(float v) { return v; }
__device__ __forceinline__ float io_widen(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
// f32 -> bf16, round to nearest even: v_cvt_pk_bf16_f32 (gfx950), two values per instruction
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pack2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ unsigned short bf16_rne(float f) { return (unsigned short)(bf16_pack2(f, 0.f) & 0xffffu); }
struct __attribute__((packed, aligned(2))) h8u { unsigned w[4]; };      // eight bf16, 2-byte aligned
__device__ __forceinline__ void io_store4(float* dst, float a, float b, float c, float d) {
    *reinterpret_cast<f4u*>(dst) = f4u{{a, b, c, d}};
}
__device__ __forceinline__ void io_store4(unsigned short* dst, float a, float b, float c, float d) {
    struct __attribute__((packed, aligned(2))) h4w { unsigned w[2]; };
    *reinterpret_cast<h4w*>(dst) = h4w{{bf16_pack2(a, b), bf16_pack2(c, d)}};
}
__device__ __forceinline__ void io_store8(unsigned short* dst, const f32x4& a, const f32x4& b) {
    *reinterpret_cast<h8u*>(dst) = h8u{{bf16_pack2(a[0], a[1]), bf16_pack2(a[2], a[3]), bf16_pack2(b[0], b[1]), bf16_pack2(b[2], b[3])}};
}
// four consecutive elements at ELEMENT alignment (rows of odd length), widened to f32
__device__ __forceinline__ float4 io_ld4(const float* p) {
    const f4u v = *reinterpret_cast<const f4u*>(p);
    return make_float4(v.v[0], v.v[1], v.v[2], v.v[3]);
}
__device__ __forceinline__ float4 io_ld4(const unsigned short* p) {
    const h4u v = *reinterpret_cast<const h4u*>(p);
    return make_float4(io_widen(v.v[0]), io_widen(v.v[1]), io_widen(v.v[2]), io_widen(v.v[3]));
}
__device__ __forceinline__ void io_store1(float* dst, float a) { *dst = a; }
__device__ __forceinline__ void io_store1(unsigned short* dst, float a) { *dst = bf16_rne(a); }

__device__ __forceinline__ float2 lds_tw(const float2* tab, unsigned byte_off) {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(tab) + byte_off);
}

// Hermitian weight of column l of a one-sided spectrum of a length-N real axis.
__device__ __forceinline__ float herm_weight(int l, int N) {
    return (l == 0 || 2 * l == N) ? 1.0f : 2.0f;
}

// Spectrum row of corner-row index j (0 <= j < 2m) on a full-complex axis of length N:
// the "lo" corner holds rows 0..m-1, the "hi" corner rows N-m..N-1.
__host__ __device__ __forceinline__ int corner_freq(int j, int m, int N) { return j < m ? j : N - 2 * m + j; }

// "Later slice-assignment wins": lo-corner row j is overwritten by the hi corner when j >= N - m.
__host__ __device__ __forceinline__ bool row_survives(int j, int m, int N) { return j >= m || j < N - m; }

// Waves per image (workgroup size / 64) for the row-tiled DFT kernels: the 16-row tiles of an image are
// dealt round-robin to the waves, so pick the count in {4, 3, 2, 1} that leaves the fewest idle tile slots
// (421 rows = 27 tiles -> 3 waves x 9; 446 rows = 28 tiles -> 4 x 7), preferring more waves on a tie.
inline int pick_waves_per_image(int n_row_tiles) {
    int best = 1, best_waste = 1 << 30;
    for (int nw = 4; nw >= 1; --nw) {
        if (nw > n_row_tiles) continue;
        const int per = (n_row_tiles + nw - 1) / nw;
        const int waste = (per * nw - n_row_tiles) * 12 / nw;      // idle slots, normalised to 12 waves
        if (waste < best_waste) { best_waste = waste; best = nw; }
    }
    return best;
}

__device__ __forceinline__ int sweep_x(int rev) { return rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x; }
__device__ __forceinline__ int sweep_y(int rev) { return rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y; }

struct Dft2dParams {
    const float* in;        // forward: images (n_img, H, W) f32; inverse: spectra (n_img, 2*m1, m2) c64
    float* out;             // forward: spectra; inverse: images
    const float2* twH;      // (cos, sin)(2 pi n / H), n in [0, H)
    const float2* twW;
    int n_img, H, W, m1, m2;
    float scale;            // applied to every spectrum entry
    int herm;               // 1: multiply column l by the Hermitian weight c_l of W
    int mask;               // 1: zero lo-corner rows overwritten by the hi corner (later-wins)
    // spectrum of image i lives at index (i / sp_group) * sp_stride + sp_offset + i % sp_group: a (B, C1) batch of
    // images can face channels [sp_offset, sp_offset + C1) of a (B, sp_stride) spectrum tensor (two-source blocks)
    int sp_group, sp_stride, sp_offset;
    int bf16;               // 1: the images (forward input / inverse output) are bfloat16; spectra stay complex64
    const int* rowfreq;     // optional (plane-batched kernels only): frequency of spectrum row j, j < 2*m1, instead of the corner rule
    int nw;                 // set by the K1 / K3 launchers: waves per image (a workgroup holds blockDim / (64 nw) images)
    int exp;                // development: knock-out switches of the bf16-MFMA kernels (dft2d_b16.hip), 0 in production
    int accumulate;         // plane-batched inverse only: out += result (the point-wise branch of OperatorBlock_3D lands in the spectral branch's buffer)
    float* act_out;         // ... and, if set, act_out = gelu(out) is written in the same pass (blocks without normalisation)
    // K3-A (dft2d_inv_add_kernel.h): out = transform + separable banded up-sampling of add_src (n_img, add_Hs, add_Ws); host-built operand
    // tables (uno_amd/resample.py): first source row of each 16-row tile, row operator [tile][3][64], first source column of each
    // (column tile, side), column operator [column tile][2][3][64]
    int rev = 0;            // set by the K1 / K3 launchers (row-tiled forms): workgroups walk the images in descending order
    const float* add_src = nullptr;
    int add_Hs = 0, add_Ws = 0;
    const int* add_p0 = nullptr;
    const float* add_rowop = nullptr;
    const int* add_v0 = nullptr;
    const float* add_colop = nullptr;
};

__device__ __forceinline__ size_t spectrum_index(const Dft2dParams& p, int img) {
    return (size_t)(img / p.sp_group) * p.sp_stride + p.sp_offset + img % p.sp_group;
}

// Batched per-mode complex GEMM: out(m, n, p) = sum_k A'(m, k, p) * B'(k, n, p), ' = optional conj.
// All offsets are in complex (float2) elements.  Modes are split in `ncorner` contiguous runs
// of Mc modes; operand X (X = A, B, O) of corner c starts at X.base[c].
struct ModeOperand {
    const float2* base[4];
    long long s0, s1;       // A: (m, k); B: (k, n); out: (m, n)
    int conj;
    int half;               // operand B only: elements are half-precision (re, im) pairs instead of complex64
};
struct ModeGemmParams {
    ModeOperand A, B;
    float2* out[4];
    long long o_sm, o_sn;
    int M, N, K, ncorner, Mc;
    int accumulate;         // out += instead of out = (weight gradients written straight into a parameter's gradient buffer)
};

// Pruned complex DFT along the leading axis of (n_img, H, C) <-> corner-major (n_img, 4, m1, m2, m3).
struct CdftParams {
    const float* in;
    float* out;
    const float2* tw;       // twiddles of length H
    int n_img, H, C, m1, m2, m3;
    float scale;
    int mask;               // zero lo-corner rows overwritten by the hi corner (on the spectrum side)
    const int* rowfreq;     // optional: frequency of spectrum row j, j < 2*m1, instead of the corner rule
};

// One workgroup per (sample, channel) volume: all three pruned transforms of the 3-D layer (dft3d_volume.hip).
struct Vol3dParams {
    const float* in;        // forward: volumes (n_vol, D1, D2, D3) f32; inverse: corner-major spectra (n_vol, 4, m1, m2, m3) c64
    float* out;
    const float2* tw1;      // (cos, sin)(2 pi n / (2 D1)), n in [0, 2 D1): half-shifted frequencies and the e^{i pi h / D1} twist
    const float2* tw2;      // ... of 2 D2
    const float2* tw3;      // (cos, sin)(2 pi n / D3)
    int n_vol, D1, D2, D3, m1, m2, m3;
    float scale;
    int herm;               // 1: multiply T-mode l by the Hermitian weight c_l of D3
    const float* ctab;      // host-built operand table of the kernel (filled in by the launcher)
};

const float2* twiddle_table(int N);      // device-resident, cached per (device, N); nullptr on failure
// A fresh device allocation holding `bytes` bytes of host data (the cached operand tables), or nullptr.  Safe while a stream of this
// thread - or, under the global capture mode PyTorch uses, of any thread - is being captured into a hipGraph: the allocation and the
// copy are made under the relaxed capture mode on a stream of their own and are complete on return, so a shape first seen inside a
// capture gets its tables without ending the capture (the launches that use them are captured as usual).
void* upload_table(const void* host, size_t bytes);
float2 twiddle_value(long long n, int N);      // host: (cos, sin)(2 pi n / N) as the tables hold it (f32 from f64, exact at multiples of pi / 2)

// Raise a kernel's dynamic-LDS limit to the largest size any launch of it (on this device) has asked for so far.  The driver call is
// made only when the request grows: it costs tens of microseconds, and the transforms are launched thousands of times per step.
// `slot` is a per-kernel-instantiation static array (one entry per device, zero-initialised).
inline bool ensure_dynamic_lds(const void* kernel, size_t bytes, int* slot) {
    if (bytes <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if ((size_t)__atomic_load_n(&slot[dev], __ATOMIC_RELAXED) >= bytes) return true;
    const int want = 160 * 1024;            // the hardware limit: one call covers every later size
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want) != hipSuccess) return false;
    __atomic_store_n(&slot[dev], want, __ATOMIC_RELAXED);
    return true;
}
void set_error(const char* fmt, ...);
// Compute units the launch geometries may count on: the device's, minus those the caller set aside for concurrently running
// communication kernels (uno_reserve_cus: RCCL's all-reduce of the previous gradient buckets runs beside the backward pass under data
// parallelism, and a geometry tuned to "one workgroup per CU on all CUs" would run a second round for the CUs RCCL holds)
int reserved_cus();
// Alternating sweep direction (round 6).  A tensor larger than the 256 MB Infinity Cache that one kernel writes (or reads) front to back
// leaves its TAIL in that cache; the next kernel that streams it front to back again starts with what has been evicted and evicts the
// tail before reaching it.  So consecutive launches of the streaming kernels walk their work items (images, batch entries, pixel tiles)
// in opposite directions: every launcher takes the next direction from a per-thread counter.  Only the ORDER changes - workgroup k
// computes work item G - 1 - k, writes that item's outputs and partial-sum slots - never the result.  uno_sweep_alternation(0) turns
// it off (every launch front to back).
int next_sweep_reversed(int family);       // family: SWEEP_* bit (uno_sweep_alternation takes a mask of them; 1 = all)
enum { SWEEP_K1 = 1, SWEEP_K3 = 2, SWEEP_K7 = 4, SWEEP_K8 = 8, SWEEP_K9 = 16, SWEEP_NORM = 32, SWEEP_PROJ = 64, SWEEP_LIFT = 128 };
inline int usable_cus(int device_cus) { const int r = reserved_cus(); return device_cus - r >= 8 ? device_cus - r : (device_cus < 8 ? device_cus : 8); }

// RAII timing scope around one kernel launch (no-op unless uno_profile_begin() is active).
struct ProfScope {
    int slot;
    hipStream_t stream;
    ProfScope(const char* name, double bytes, hipStream_t s);
    ~ProfScope();
};

int launch_dft2d_fwd(const Dft2dParams& p, hipStream_t s);
int launch_dft2d_inv(const Dft2dParams& p, hipStream_t s);
bool dft2d_inv_add_applies(const Dft2dParams& p);       // dft2d_inv_add.hip: K3 + up-sampled addend (p.add_* set)
int launch_dft2d_inv_add(const Dft2dParams& p, hipStream_t s);
bool dft2d_fwd_plane_applies(const Dft2dParams& p);      // dft2d_plane.hip: many small images
bool dft2d_inv_plane_applies(const Dft2dParams& p);
bool dft2d_b16_applies(const Dft2dParams& p);           // dft2d_b16.hip: bfloat16 images, row stage on the bf16 MFMA
int launch_dft2d_fwd_b16(const Dft2dParams& p, hipStream_t s);
int launch_dft2d_inv_b16(const Dft2dParams& p, hipStream_t s);
int launch_dft2d_fwd_plane(const Dft2dParams& p, hipStream_t s);
int launch_dft2d_inv_plane(const Dft2dParams& p, hipStream_t s);
bool vol3d_fwd_applies(int n_vol, int D1, int D2, int D3, int m1, int m2, int m3);      // dft3d_volume.hip
bool vol3d_inv_applies(int n_vol, int D1, int D2, int D3, int m1, int m2, int m3);
int launch_dft3d_fwd_volume(const Vol3dParams& p, hipStream_t s);
int launch_dft3d_inv_volume(const Vol3dParams& p, hipStream_t s);
int launch_mode_gemm(const ModeGemmParams& p, hipStream_t s);
int launch_mode_gemm_pair(const ModeGemmParams& input_grad, const ModeGemmParams& weight_grad, hipStream_t s);      // both GEMMs of a backward pass, one launch where possible
int launch_cdft(const CdftParams& p, bool inverse, hipStream_t s);
int launch_dft2d_generic(const Dft2dParams& p, bool inverse, void* ws, size_t ws_bytes, hipStream_t s);      // dft_generic.hip: any mode count; ws: 8 n_img H m2 bytes
int launch_cdft_generic(const CdftParams& p, bool inverse, hipStream_t s);
int launch_resample2d(const void* in, void* out, float* tmp, int n_img, int H, int W, int Ho, int Wo, const int* startH,
                      const float* wtH, int KH, const int* startW, const float* wtW, int KW, const int* tile_p0,
                      const float* tile_w, int NP, int accumulate, int bf16, hipStream_t s);
// K8 arguments in full: two sources (input channels [0, C1) from x, [C1, Ci) from x2), two destinations (output channels
// [0, Co1) to y, [Co1, Co) to y2; dgelu_of goes with y), optional activated copy y_act = gelu(y); nullptr x2 / y2 / y_act = plain
struct ChannelMixArgs {
    const void* x; const void* x2; const float* w; const float* bias; void* y; void* y2; void* y_act; const void* dgelu_of;
    const float* proj_w; const float* proj_b; void* proj_out;       // fused one-channel projection of gelu(y) (Co <= 64), or nullptr
    int B, Ci, Co, C1, Co1; long long P; int transpose_w, accumulate, act_in, bf16;
    PixelWindow win;                                                // all operands on one window (proj_out: one plane per batch entry)
    void* ws = nullptr; size_t ws_bytes = 0;                       // optional scratch (uno_scratch_provide): 6 Ci Co bytes let K8-S run on pre-split weights
    const float* vh_x = nullptr; const float* vh_w = nullptr; const float* vh_b = nullptr; int vh_ci = 0, vh_mode = 0;   // virtual operand (channel_mix.hip): 1 = the input, 2 = dgelu_of
    const void* gmul = nullptr;                                     // y = gelu'(product + bias) * gmul, gmul on the padded planes described below
    const float* pb_w2 = nullptr; const float* pb_g = nullptr;      // the X operand is pb_w2[k] gelu'(x[b][k][q]) pb_g[b][q] (channel_mix.hip, wide kernel)
    int act_cols = 0, act_pitch = 0; long long act_plane = 0;       // y_act on padded planes (generic kernel): the P = H * act_cols dense
                                                                    // pixels land in the top-left corner of act_plane / act_pitch rows
};
int launch_clear_border(float* t, long long n_planes, int Hp, int Wp, int rows, int cols, hipStream_t s);       // pointwise_fused.hip
int launch_channel_mix2(const ChannelMixArgs& a, hipStream_t s);
int launch_channel_mix(const void* x, const float* w, const float* bias, void* y, int B, int Ci, int Co, long long P,
                       int transpose_w, int accumulate, int act_in, const void* dgelu_of, int bf16, hipStream_t s, void* ws = nullptr,
                       size_t ws_bytes = 0);
long long channel_mix_ws_bytes(int Ci, int Co, long long P, int bf16);       // scratch that lets a call of this shape use pre-split weights (0: none)
int launch_adam_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* n,
                      const int* is_complex, double lr, double beta1, double beta2, double eps, double wd, int step, hipStream_t s,
                      const float* dev_scalars = nullptr);
int launch_adam_advance(int* step, float* scalars, const double* hyper, double lr, double eps, double wd, double beta1, double beta2,
                        hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, long long n, int is_complex, double lr, double beta1, double beta2,
                double eps, double wd, int step, hipStream_t s);
int launch_gelu_project_fwd(const void* pre, const float* w, const float* bias, void* out, int B, int C, long long P, int bf16, hipStream_t s);
long long gelu_project_ws_floats(int B, int C, long long P);
int launch_gelu_project_bwd(const void* pre, const float* w, const void* gout, void* gpre, float* gw, float* gb, float* ws, int B,
                            int C, long long P, int bf16, hipStream_t s, const PixelWindow& win = PixelWindow());
int launch_gelu_pad(const void* s, const void* gy, void* out, int n_img, int H, int W, int Hp, int Wp, int backward, int bf16, hipStream_t st);
int launch_transpose_batched(const float* in, float* out, int B, long long R, int C, long long ld_in, long long sb_in, long long ld_out,
                             long long sb_out, hipStream_t st);
int launch_instnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long long rows, int C,
                        long long N, float eps, int gelu, int bf16, hipStream_t s);
int launch_instnorm_bwd(const void* x, const void* gy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        void* gx, float* s1, float* s2, long long rows, int C, long long N, int gelu, int bf16, hipStream_t s);
long long channel_wgrad_ws_floats(int B, int Ci, int Co, long long P, int* nsplit_out);
int launch_channel_wgrad(const void* gy, const void* x, float* gw, float* gb, float* ws, int B, int Ci, int Co, long long P,
                         int act_x, int bf16, hipStream_t s);
// pb (channel_mix.hip, ChannelWgradParams::pb_*): gy is the layer's pre-activation and stands for w2[o] gelu'(gy) g; the projection's own
// gradients go to gw2 (Co) / gb2 (1, may be null); ws then holds channel_wgrad_pb_ws_floats() floats
struct WgradProjectedBack { const float* w2 = nullptr; const float* g = nullptr; float* gw2 = nullptr; float* gb2 = nullptr; };
bool channel_wgrad_pb_applies(int B, int Ci, int Co, int C1, long long P);
long long channel_wgrad_pb_ws_floats(int B, int Ci, int Co, long long P);
int launch_channel_wgrad2(const void* gy, const void* x, const void* x2, int C1, float* gw, float* gb, float* ws, int B, int Ci, int Co,
                          long long P, int act_x, int accumulate, int bf16, hipStream_t s, const PixelWindow& win = PixelWindow(),
                          const WgradProjectedBack& pb = WgradProjectedBack());
// weight gradient with the X operand virtual (see ChannelMixParams: gelu of it is taken when act_x): Ci <= 32 virtual channels
int launch_channel_wgrad_vh(const void* gy, const float* vh_x, const float* vh_w, const float* vh_b, int vh_ci, float* gw, float* gb, float* ws,
                            int B, int Ci, int Co, long long P, int act_x, hipStream_t s, int accumulate = 0);
// lift_bwd.hip: the lift's backward pass with gz in LDS only (32 middle / 64 output channels)
bool lift_bwd_fused_applies(int Cin, int Cm, int Co, int W, long long P);
long long lift_bwd_fused_parts(int B, int H, int W);
int launch_lift_forward_fused(const float* x, const float* w1, const float* b1, const float* w0, const float* b0, float* act, int B, int Cin,
                              int H, int W, int Hp, int Wp, hipStream_t s);       // writes rows 0 .. H - 1 of the padded planes in full
int launch_lift_backward_fused(const float* x, const float* w1, const float* b1, const float* w0, const float* b0, const float* g, float* part,
                               float* part1, int B, int Cin, int H, int W, int Hp, int Wp, hipStream_t s, const float* g2 = nullptr);
int launch_channel_wgrad_finish(const float* parts, float* gw, float* gb, int Ci, int Co, long long nparts, int accumulate, hipStream_t s);

}  // namespace uno
