// K8 / K9 - channel mixing of channels-first tensors: the 1x1 convolution of pointwise_op_2D/3D (reference
// integral_operators.py:210-243, 430-468: nn.Conv2d/3d(in, out, 1)) and the lift / projection Linear layers of the
// U-NO models applied channels-first.
//
//   K8  Y[b][o][p] = sum_i Wm(o, i) * X[b][i][p] (+ bias[o])      forward, and - with Wm = W^T - the input gradient
//   K9  gW[o][i]   = sum_{b,p} gY[b][o][p] * X[b][i][p],  gb[o] = sum_{b,p} gY[b][o][p]      weight / bias gradient
//
// X, Y are (B, C, P) with the pixel axis contiguous (no layout change for NCHW tensors).  rocBLAS runs these
// skinny shapes (M = K = 64..256 channels, N = 12k..200k pixels) at ~30 % of what their memory traffic allows;
// here both operands are staged through LDS with unit-stride loads along the pixel axis and consumed as
// v_mfma_f32_16x16x4_f32 fragments (exact f32, same arithmetic as an fmaf chain).
#include "uno_common.h"
#include <cstdio>
#include <cstdlib>

namespace uno {

// ------------------------------------------------------------------------------------------------ K8
constexpr int CM_PT = 128;          // pixels per workgroup
constexpr int CM_MT = 64;           // output channels per workgroup (4 waves x 16)
constexpr int CM_KC = 16;           // input channels per staged chunk
constexpr int CM_WS = CM_MT + 16;   // LDS row stride of the W chunk  [KC][MT] (k-major)

// four consecutive floats starting at row[px] with zeros past the row end (row has P >= 4 valid floats):
// one unconditional 16-byte load from a clamped address, then a select network (no branches -> the load
// stays in flight across the MFMA block)
template <typename T>
__device__ __forceinline__ float4 load4_tail(const T* row, int px, int P, const PixRun& run = PixRun{0, 0x7fffffff, 0}) {
    const int pc = min(px, P - 4);
    const float4 v = io_ld4(row + run(pc));
    const int sh = px - pc;
    float t0 = v.x, t1 = v.y, t2 = v.z, t3 = v.w;
    if (sh & 1) { t0 = t1; t1 = t2; t2 = t3; t3 = 0.f; }
    if (sh & 2) { t0 = t2; t1 = t3; t2 = 0.f; t3 = 0.f; }
    if (sh >= 4) { t0 = 0.f; t1 = 0.f; t2 = 0.f; t3 = 0.f; }
    return make_float4(t0, t1, t2, t3);
}

// exact-erf GELU (F.gelu default) and its derivative, for the fused forms: x := gelu(x) while a chunk goes to LDS
// (`act_in`: the layer consumes the activation of a tensor that is kept pre-activation) and y := (W x) * gelu'(pre) in the
// epilogue (`dgelu_of`: the input gradient of such a layer, handed back as the gradient of the pre-activation tensor)
__device__ __forceinline__ float cm_gelu(float x) { return uno_gelu(x); }
__device__ __forceinline__ float cm_dgelu(float x) { return uno_dgelu(x); }
__device__ __forceinline__ float4 cm_gelu4(float4 v) { return make_float4(cm_gelu(v.x), cm_gelu(v.y), cm_gelu(v.z), cm_gelu(v.w)); }
// four consecutive logical pixels px .. px + 3 through a run map whose rows need not be a multiple of 4 long
template <typename T>
__device__ __forceinline__ float4 load4_run(const T* plane, const PixRun& run, int px) {
    const int f0 = run(px), f3 = run(px + 3);
    if (f3 - f0 == 3) return io_ld4(plane + f0);
    return make_float4(io_widen(plane[f0]), io_widen(plane[run(px + 1)]), io_widen(plane[run(px + 2)]), io_widen(plane[f3]));
}
template <typename T>
__device__ __forceinline__ void store4_run(T* plane, const PixRun& run, int px, float a, float b, float c, float d) {
    const int f0 = run(px), f3 = run(px + 3);
    if (f3 - f0 == 3) { io_store4(plane + f0, a, b, c, d); return; }
    io_store1(plane + f0, a); io_store1(plane + run(px + 1), b); io_store1(plane + run(px + 2), c); io_store1(plane + f3, d);
}

struct ChannelMixParams {
    const void* x;          // (B, C1, P) f32 | bf16 (BF instantiations: activations bfloat16, weights / accumulation f32)
    const void* x2;         // (B, Ci - C1, P): input channels [C1, Ci) of a two-source call (a skip connection's torch.cat that is
                            // never built), or nullptr with C1 == Ci
    const float* w;         // Wm(o, i) = w[o * w_so + i * w_si]
    const float* bias;      // (Co) or nullptr
    void* y;                // (B, Co1, P), x's element type
    void* y2;               // (B, Co - Co1, P): output channels [Co1, Co) of a two-destination call (the two input gradients of a
                            // two-source layer from one pass over grad_y), or nullptr with Co1 == Co
    void* y_act;            // nullptr, or (B, Co, P): also receives gelu(y) (the block's activation written by the kernel that
                            // completes the pre-activation sum; single destination only)
    const float* proj_w;    // nullptr, or (Co): the kernel also writes proj_out[b][p] = proj_b + sum_o proj_w[o] gelu(y[b][o][p]) - the
    const float* proj_b;    // models' final `fc2(F.gelu(fc1(x)))` with one output channel (darcy_flow_uno2d.py:128-131) without
    void* proj_out;         // re-reading y; needs all output channels in ONE 64-channel tile (Co <= 64)
    int B, Ci, Co, P;
    PixMap pm;              // plane stride of every operand + the pixel window (dense: pm.PS == P); generic and split kernels only
    const void* gmul;       // nullptr, or (B, Co, padded plane) read through pm_act: y = gelu'(product + bias) * gmul - the gradient of the lift's
                            // last pre-activation RECOMPUTED from the layer's input instead of stored (generic kernel, one destination)
    // VIRTUAL operand (generic kernel, template VH): a (B, nvh <= 64, P) tensor that is never stored - channel c at pixel q is
    // vh_b[c] + sum_k vh_w[c][k] vh_x[b][k][q] with vh_ci <= 3 real channels (the lift's first layer, reference darcy_flow_uno2d.py:98:
    // `fc_n1` on [a(x, y), x, y]; 12 bytes per pixel instead of 128).  VH = 1: it is the X operand (Ci channels); VH = 2: it is dgelu_of
    const float* vh_x; const float* vh_w; const float* vh_b; int vh_ci;
    int store_y;            // 0: only y_act is written (the padded-activation call that does not keep the pre-activation result)
    PixMap pm_act;          // y_act's OWN map when pm_act.rl != 0 (generic kernel): the dense pixels of an H x W grid go to the top-left
                            // corner of (Hp, Wp) planes - rl = W (any width: a lane's four pixels may straddle a row end), skip = Wp - W
    int C1, Co1;
    int w_so, w_si;
    int rev;                // alternating sweep direction: pixel tiles and batch entries in descending order
    int ncot;               // channel tiles per pixel tile
    int ntile, per_xcd;     // tiles (pixel x channel) per batch entry; ceil(ntile / 8)
    const void* wsplit;     // K8-S with WM = 2: the weights as bf16 pieces in the kernel's own LDS layout (channel_mix_wsplit_kernel)
    int exp;                // development knock-outs of K8-S (timing only; UNO_CMS_EXP)
    int accumulate;         // 1: y += instead of y =;  2 (with dgelu_of): y = (y + product) * gelu'(dgelu_of) - the LAST contribution to a
                            // gradient that must still pass through a GELU: the factor multiplies the completed sum
    const void* dgelu_of;   // nullptr, or (B, Co1, P): the product is multiplied by gelu'(dgelu_of) before bias-free accumulation
                            // (first destination only)
    // PROJECTED-BACK operand (wide kernel, template PB; round 6): the X operand of the call is the gradient at the output of the layer
    // BEFORE a one-channel projection, never stored - x[b][k][q] stands for pb_w2[k] gelu'(x[b][k][q]) pb_g[b][q] with x the layer's
    // pre-activation, pb_w2 (Ci) the projection's weights and pb_g (B, plane) the gradient at the projection's output (reference
    // darcy_flow_uno2d.py:128-131 backward: `fc2(F.gelu(fc1(x)))`).  The kernel applies gelu' as it stages x, pb_w2 as it stages the
    // weights and pb_g to the finished product.
    const float* pb_w2; const float* pb_g;
};

// where output channel tile o0 of batch entry b lives: row o of the tile = base + (o - ob) * P
template <typename T>
struct CmDest { T* base; int ob; };
template <typename T>
__device__ __forceinline__ CmDest<T> cm_dest(const ChannelMixParams& p, int o0, int b) {
    if (o0 >= p.Co1) return {reinterpret_cast<T*>(p.y2) + (size_t)b * (p.Co - p.Co1) * p.pm.PS, p.Co1};
    return {reinterpret_cast<T*>(p.y) + (size_t)b * p.Co1 * p.pm.PS, 0};
}

// MODE 2: interior tile (128 whole pixels, 64 whole output channels, input channels a multiple of 16): no guards,
//         32-bit offsets from a uniform base - the per-element clamps and selects of the guarded path cost more
//         VALU issue slots than the tile has MFMAs;  MODE 1: guarded 16-byte loads (P >= 4);  MODE 0: guarded scalars.
template <int MODE, int PT, bool ACT = false, bool DG = false, bool BF = false, int VH = 0>
__device__ __forceinline__ void channel_mix_tile(const ChannelMixParams& p, float (*sX)[CM_KC * (PT + 16)], float (*sW)[CM_KC * CM_WS],
                                                 int p0, int o0, int b, const float4* sVH = nullptr) {
    static_assert(VH == 0 || (MODE != 0 && !BF), "virtual operands: vector modes, float32");
    constexpr int XS = PT + 16;         // LDS row stride of the X chunk [KC][PT]: 4 consecutive rows hit disjoint bank groups
    constexpr int F4R = PT / 4;         // 16-byte pieces per row
    constexpr int NV = PT / 64;         // 16-byte pieces per thread per chunk
    constexpr int NM = PT / 16;         // pixel tiles of 16 = accumulators per wave
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    using T = typename IoElem<BF>::type;                // float | unsigned short (bf16 bits)
    const int C1 = p.C1;
    const int PS = p.pm.PS;                      // elements between two channel planes (== P without a window)
    const PixRun run = pix_run(p.pm, p0);        // logical pixel of this tile -> offset inside its plane
    const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * C1 * PS;                        // channels [0, C1)
    const T* xb2 = p.x2 ? reinterpret_cast<const T*>(p.x2) + (size_t)b * (p.Ci - C1) * PS : xb;  // channels [C1, Ci), row ci - C1
    const CmDest<T> dd = cm_dest<T>(p, o0, b);
    T* const ydst = dd.base;                     // row o of this tile: ydst + (o - dd.ob) * P
    const bool dg = DG && o0 < p.Co1;            // gelu' factor: first destination only
    const T* const dall = DG ? reinterpret_cast<const T*>(p.dgelu_of) + (size_t)b * p.Co1 * PS : nullptr;
    const bool act_map = p.pm_act.rl != 0;       // y_act on its own (padded) planes
    const int APS = act_map ? p.pm_act.PS : PS;
    const PixRun arun = act_map ? pix_run(p.pm_act, p0) : run;
    T* const aall = p.y_act ? reinterpret_cast<T*>(p.y_act) + (size_t)b * p.Co * APS : nullptr;
    const T* const gall = p.gmul ? reinterpret_cast<const T*>(p.gmul) + (size_t)b * p.Co * APS : nullptr;
    const bool store_y = p.store_y != 0;
    bool act_ld = ACT;                           // the chunk in the staging registers comes from the activated source
    // virtual operand: this thread's four pixels (tile pixel 4 (tid % 32): the staging pixel AND the interior epilogue's) of the real channels
    float4 vx[3];
    if constexpr (VH != 0) {
        const float* vb = p.vh_x + (size_t)b * p.vh_ci * p.P;
        const int vpx = p0 + (tid & (F4R - 1)) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* row = vb + (size_t)min(k, p.vh_ci - 1) * p.P;
            vx[k] = MODE == 2 ? io_ld4(row + vpx) : load4_tail(row, vpx, p.P);
            if (k >= p.vh_ci) vx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    auto vh_of = [&](int c, const float4* q) {  // channel c of the virtual tensor at the four pixels whose real channels are q[0..2]
        const float4 t = sVH[c];                 // (w[c][0], w[c][1], w[c][2], b[c])
        return make_float4(fmaf(t.z, q[2].x, fmaf(t.y, q[1].x, fmaf(t.x, q[0].x, t.w))), fmaf(t.z, q[2].y, fmaf(t.y, q[1].y, fmaf(t.x, q[0].y, t.w))),
                           fmaf(t.z, q[2].z, fmaf(t.y, q[1].z, fmaf(t.x, q[0].z, t.w))), fmaf(t.z, q[2].w, fmaf(t.y, q[1].w, fmaf(t.x, q[0].w, t.w))));
    };
    auto vh_at = [&](int c, int px) {            // ... at any pixel quad of the tile (guarded loads; the edge tile's epilogue)
        const float* vb = p.vh_x + (size_t)b * p.vh_ci * p.P;
        float4 q[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            q[k] = load4_tail(vb + (size_t)min(k, p.vh_ci - 1) * p.P, px, p.P);
            if (k >= p.vh_ci) q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return vh_of(c, q);
    };

    // staging maps: X chunk = 16 rows x 128 px -> two 16-byte pieces per thread (row e / 32, px 4 (e % 32)) or, MODE 0,
    //               8 single elements (row e / 128, px e % 128);  W chunk = 16 k x 64 o -> 4 elements (k e % 16, o e / 16)
    float4 rx[2];      // NV used in the vector modes, all 8 floats in MODE 0
    float rw[4];
    const bool tr = p.w_so == 1 && p.w_si != 1;        // W contiguous along the output channel (input-gradient call)
    auto load_chunk = [&](int k0) {
        act_ld = ACT && k0 < C1;
        if constexpr (MODE == 2) {
            // a chunk of 16 channels lies in one source (C1 is a multiple of 16 in two-source calls)
            const T* cb = k0 < C1 ? xb : xb2;
            const int kb = k0 < C1 ? k0 : k0 - C1;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = tid + 256 * u;
                if constexpr (VH == 1) rx[u] = vh_of(k0 + e / F4R, vx);
                else rx[u] = io_ld4(cb + (unsigned)((kb + (e / F4R)) * PS + run(p0 + (e % F4R) * 4)));
            }
            // W chunk as ONE 16-byte load per thread along whichever index is contiguous in memory (4 dword loads
            // per thread made W the most numerous vector-memory instruction of the tile; the address unit was the
            // busiest block of the CU)
            // (rows / columns of W past the last output channel of a partial tile are clamped: their products are never stored)
            const unsigned woff = tr ? (unsigned)((k0 + (tid >> 4)) * p.w_si + min(o0 + (tid & 15) * 4, p.Co - 4))
                                     : (unsigned)(min(o0 + (tid >> 2), p.Co - 1) * p.w_so + k0 + (tid & 3) * 4);
            const f4u wv = *reinterpret_cast<const f4u*>(p.w + woff);
#pragma unroll
            for (int j = 0; j < 4; ++j) rw[j] = wv.v[j];
            return;
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = tid + 256 * u;
                const int ci = k0 + (e / F4R), cc = min(ci, p.Ci - 1);
                float4 v;
                if constexpr (VH == 1) v = vh_of(cc, vx);
                else v = load4_tail(cc < C1 ? xb + (size_t)cc * PS : xb2 + (size_t)(cc - C1) * PS, p0 + (e % F4R) * 4, p.P, run);
                if (ci >= p.Ci) v = make_float4(0.f, 0.f, 0.f, 0.f);
                rx[u] = v;
            }
        }
        if constexpr (MODE == 0) {
            float* r = reinterpret_cast<float*>(rx);
#pragma unroll
            for (int u = 0; u < PT / 16; ++u) {
                const int e = tid + 256 * u;
                const int ci = k0 + e / PT, pp = p0 + e % PT;
                r[u] = (ci < p.Ci && pp < p.P) ? io_widen(ci < C1 ? xb[(size_t)ci * PS + run(pp)] : xb2[(size_t)(ci - C1) * PS + run(pp)]) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            const int k = e & 15, o = e >> 4;
            const int ci = k0 + k, oo = o0 + o;
            rw[u] = (ci < p.Ci && oo < p.Co) ? p.w[oo * p.w_so + ci * p.w_si] : 0.f;
        }
    };
    auto store_chunk = [&](int buf) {
        if constexpr (MODE != 0) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = tid + 256 * u;
                *reinterpret_cast<float4*>(&sX[buf][(e / F4R) * XS + (e % F4R) * 4]) = (ACT && act_ld) ? cm_gelu4(rx[u]) : rx[u];
            }
        } else {
            const float* r = reinterpret_cast<const float*>(rx);
#pragma unroll
            for (int u = 0; u < PT / 16; ++u) {
                const int e = tid + 256 * u;
                sX[buf][(e / PT) * XS + e % PT] = (ACT && act_ld) ? cm_gelu(r[u]) : r[u];
            }
        }
        if constexpr (MODE == 2) {
            float* d = sW[buf] + (tr ? (tid >> 4) * CM_WS + (tid & 15) * 4 : (tid & 3) * 4 * CM_WS + (tid >> 2));
            const int st = tr ? 1 : CM_WS;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j * st] = rw[j];
            return;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            sW[buf][(e & 15) * CM_WS + (e >> 4)] = rw[u];
        }
    };

    f32x4 acc[NM];
#pragma unroll
    for (int nt = 0; nt < NM; ++nt) acc[nt] = f32x4{0, 0, 0, 0};

    const int nchunk = (p.Ci + CM_KC - 1) / CM_KC;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    const bool wave_ok = MODE != 2 || o0 + 16 * wave < p.Co;         // interior path of a partial channel tile: whole waves idle
    // ... unless the tile has at most 32 valid output channels (the 32-channel input gradient of the lift): then waves 2, 3 would idle
    // through the whole K loop while waves 0, 1 - two SIMDs of four - do all the MFMAs.  The K steps of a chunk are split instead: wave
    // (c, h) takes channel group c = wave & 1 and k-steps 2 h, 2 h + 1 (h = wave >> 1); the two partial sums meet in LDS once per tile.
    const bool ksplit = MODE == 2 && PT == 128 && o0 + 32 >= p.Co;
    const int wc = ksplit ? (wave & 1) : wave;
    const bool mul_ok = ksplit ? (o0 + 16 * wc < p.Co) : wave_ok;
    const int ks_lo = ksplit ? 2 * (wave >> 1) : 0, ks_hi = ksplit ? ks_lo + 2 : CM_KC / 4;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) load_chunk((c + 1) * CM_KC);        // global -> registers while this chunk is multiplied
        if (mul_ok) {
#pragma unroll
            for (int ks = 0; ks < CM_KC / 4; ++ks) {
                if (ks < ks_lo || ks >= ks_hi) continue;                                     // (uniform)
                // D^T = X^T W^T: A[i = px][k] = X[k][16 mt + r16], B[k][j = o] = Wm(16 wave + r16, k): a lane ends up with
                // 4 consecutive pixels of one output channel -> one 16-byte store
                const float wv = sW[buf][(4 * ks + kk) * CM_WS + 16 * wc + r16];
                const float* xrow = sX[buf] + (4 * ks + kk) * XS + r16;
#pragma unroll
                for (int mt = 0; mt < NM; ++mt) acc[mt] = mfma16(xrow[16 * mt], wv, acc[mt]);
            }
        }
        if (c + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }
    if (ksplit) {
        // partial sums of waves 2, 3 -> waves 0, 1 (the X staging buffers are free: 2 waves x NM tiles x 64 lanes x 16 bytes = 16 KB)
        float* sR = &sX[0][0];
        if (wave >= 2) {
#pragma unroll
            for (int mt = 0; mt < NM; ++mt) *reinterpret_cast<f32x4*>(sR + (((wave - 2) * NM + mt) * 64 + lane) * 4) = acc[mt];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int mt = 0; mt < NM; ++mt) acc[mt] += *reinterpret_cast<const f32x4*>(sR + ((wave * NM + mt) * 64 + lane) * 4);
        }
        __syncthreads();
    }

    // D[px = 16 mt + 4 kk + r][o = 16 wave + r16]
    if constexpr (MODE == 2 && PT == 128) {
        // rows of 128 pixels leave as contiguous 512-byte stores - 2 rows per instruction instead of 16 rows x 64 B,
        // a third of the cache-line requests (measured -7..13 % kernel time once W no longer dominated the address
        // unit): the wave's 16 x 128 tile goes through LDS in two halves of 8 channels (staging buffers are free now)
        constexpr int OS = PT + 16;         // 64-bank LDS: the 8 rows x 4 lane groups of a 16-byte write spread over all banks
        float* sO = &sX[0][0] + wave * (8 * OS);
        const int c4 = (lane & 31) * 4;
        // bias of the wave's 16 channels: one load per lane up front, handed out by shuffle (a load per stored row inside
        // the loop below put an L2 round trip in front of every store); likewise all reads of an accumulating call are
        // issued before the first store (the compiler must keep a later load of y behind an earlier store to y)
        const float bias_l = (p.bias && wave_ok) ? p.bias[o0 + 16 * wave + r16] : 0.f;
        const float pw_l = (p.proj_w && wave_ok) ? p.proj_w[o0 + 16 * wave + r16] : 0.f;
        float ps[4] = {0.f, 0.f, 0.f, 0.f};            // projection: this lane's 4 pixels, summed over the rows it handles
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((r16 >> 3) == h) {
#pragma unroll
                for (int mt = 0; mt < NM; ++mt)
                    *reinterpret_cast<float4*>(sO + (r16 & 7) * OS + 16 * mt + 4 * kk) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
            }
            __syncthreads();
            float4 old[4], pre[DG ? 4 : 1];
            T* dst[4];
            size_t aoff[4];
            if (wave_ok) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int o = o0 + 16 * wave + 8 * h + 2 * it + (lane >> 5);
                const size_t off = (size_t)(o - dd.ob) * PS + run(p0 + c4);
                dst[it] = ydst + off;
                aoff[it] = (size_t)o * APS;
                if (p.accumulate) old[it] = io_ld4(dst[it]);
                else old[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (DG) {
                    if constexpr (VH == 2) pre[it] = dg ? vh_of(o - dd.ob, vx) : make_float4(0.f, 0.f, 0.f, 0.f);
                    else { if (dg) pre[it] = io_ld4(dall + off); else pre[it] = make_float4(0.f, 0.f, 0.f, 0.f); }
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = 2 * it + (lane >> 5);
                const float4 v = *reinterpret_cast<const float4*>(sO + row * OS + c4);
                const float bv = __shfl(bias_l, 8 * h + row);
                float r4[4] = {v.x + bv, v.y + bv, v.z + bv, v.w + bv};
                const float o4[4] = {old[it].x, old[it].y, old[it].z, old[it].w};
                float w4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w4[i] = o4[i] + r4[i];
                    if constexpr (DG) {
                        const float pr4[4] = {pre[it].x, pre[it].y, pre[it].z, pre[it].w};
                        if (dg) { const float d = cm_dgelu(pr4[i]); w4[i] = p.accumulate == 2 ? w4[i] * d : fmaf(r4[i], d, o4[i]); }
                    }
                }
                if (gall) {
                    const float4 g4 = load4_run(gall + aoff[it], arun, p0 + c4);
                    w4[0] = cm_dgelu(w4[0]) * g4.x; w4[1] = cm_dgelu(w4[1]) * g4.y; w4[2] = cm_dgelu(w4[2]) * g4.z; w4[3] = cm_dgelu(w4[3]) * g4.w;
                }
                if (store_y) io_store4(dst[it], w4[0], w4[1], w4[2], w4[3]);
                if (aall) store4_run(aall + aoff[it], arun, p0 + c4, cm_gelu(w4[0]), cm_gelu(w4[1]), cm_gelu(w4[2]), cm_gelu(w4[3]));
                if (p.proj_w) {
                    const float pwv = __shfl(pw_l, 8 * h + row);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ps[i] = fmaf(pwv, cm_gelu(w4[i]), ps[i]);
                }
            }
            }
            __syncthreads();
        }
        if (p.proj_w) {
            // lanes l and l + 32 hold the even / odd rows of the same 4 pixels; then the four waves (16 channels each) through LDS
            // in a fixed order (the W staging buffers are free after the K loop)
            float* sP = &sW[0][0];
#pragma unroll
            for (int i = 0; i < 4; ++i) ps[i] += __shfl_xor(ps[i], 32);
            if (lane < 32) *reinterpret_cast<float4*>(sP + wave * PT + c4) = make_float4(ps[0], ps[1], ps[2], ps[3]);
            __syncthreads();
            if (wave == 0 && lane < 32) {
                const float pb = p.proj_b ? p.proj_b[0] : 0.f;
                float r4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) r4[i] = pb + (((sP[c4 + i] + sP[PT + c4 + i]) + sP[2 * PT + c4 + i]) + sP[3 * PT + c4 + i]);
                io_store4(reinterpret_cast<T*>(p.proj_out) + (size_t)b * PS + run(p0 + c4), r4[0], r4[1], r4[2], r4[3]);
            }
        }
        return;
    }
    const int o = o0 + 16 * wave + r16;
    float pv[NM][4];                            // projection terms proj_w[o] gelu(y[o][px]) of this lane (zero where nothing is stored)
#pragma unroll
    for (int mt = 0; mt < NM; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[mt][r] = 0.f;
    if (MODE == 2 || o < p.Co) {
        const float bv = p.bias ? p.bias[o] : 0.f;
        const float pwv = p.proj_w ? p.proj_w[o] : 0.f;
        T* yrow = ydst + (size_t)(o - dd.ob) * PS;
        T* arow = aall ? aall + (size_t)o * APS : nullptr;
        const T* grow = gall ? gall + (size_t)o * APS : nullptr;
#pragma unroll
        for (int mt = 0; mt < NM; ++mt) {
            const int px = p0 + 16 * mt + 4 * kk, fx = run(px);        // logical pixel (guards), offset inside the plane
            const T* drow = (dg && VH != 2) ? dall + (size_t)(o - dd.ob) * PS : nullptr;
            float4 vpre = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (DG && VH == 2) { if (dg) vpre = vh_at(o - dd.ob, px); }
            if (MODE == 2 || px + 3 < p.P) {
                float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f), pre = o4;
                if (p.accumulate) o4 = io_ld4(yrow + fx);
                if constexpr (DG) { if constexpr (VH == 2) pre = vpre; else if (dg) pre = io_ld4(drow + fx); }
                float w4[4] = {o4.x, o4.y, o4.z, o4.w};
                const float pr4[4] = {pre.x, pre.y, pre.z, pre.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = dg ? cm_dgelu(pr4[r]) : 1.f;
                    w4[r] = p.accumulate == 2 ? (w4[r] + (acc[mt][r] + bv)) * d : w4[r] + (acc[mt][r] + bv) * d;
                }
                if (grow) {
                    const float4 g4 = load4_run(grow, arun, px);
                    w4[0] = cm_dgelu(w4[0]) * g4.x; w4[1] = cm_dgelu(w4[1]) * g4.y; w4[2] = cm_dgelu(w4[2]) * g4.z; w4[3] = cm_dgelu(w4[3]) * g4.w;
                }
                if (store_y) io_store4(yrow + fx, w4[0], w4[1], w4[2], w4[3]);
                if (arow) store4_run(arow, arun, px, cm_gelu(w4[0]), cm_gelu(w4[1]), cm_gelu(w4[2]), cm_gelu(w4[3]));
                if (p.proj_w) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[mt][r] = pwv * cm_gelu(w4[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (px + r < p.P) {
                        const float vp4[4] = {vpre.x, vpre.y, vpre.z, vpre.w};
                        const float d = dg ? cm_dgelu(VH == 2 ? vp4[r] : io_widen(drow[fx + r])) : 1.f, o1 = p.accumulate ? io_widen(yrow[fx + r]) : 0.f;
                        float v = p.accumulate == 2 ? (o1 + (acc[mt][r] + bv)) * d : o1 + (acc[mt][r] + bv) * d;
                        if (grow) v = cm_dgelu(v) * io_widen(grow[arun(px + r)]);
                        if (store_y) io_store1(yrow + fx + r, v);
                        if (arow) io_store1(arow + arun(px + r), cm_gelu(v));
                        if (p.proj_w) pv[mt][r] = pwv * cm_gelu(v);
                    }
            }
        }
    }
    if (p.proj_w) {
        // sum over the wave's 16 channels (lanes r16 = 0..15 of a lane group), then the four waves through LDS in a fixed order
        float* sP = &sW[0][0];
        __syncthreads();                        // every wave is past its last read of the W staging buffer
#pragma unroll
        for (int mt = 0; mt < NM; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = pv[mt][r];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                if (r16 == 0) sP[wave * PT + 16 * mt + 4 * kk + r] = v;
            }
        __syncthreads();
        const float pb = p.proj_b ? p.proj_b[0] : 0.f;
        for (int e = tid; e < PT; e += 256)
            if (p0 + e < p.P)
                io_store1(reinterpret_cast<T*>(p.proj_out) + (size_t)b * PS + run(p0 + e), pb + (((sP[e] + sP[PT + e]) + sP[2 * PT + e]) + sP[3 * PT + e]));
    }
}

// One short-lived workgroup per (pixel tile, channel tile, batch entry).  Measured alternatives that lost:
// persistent workgroups with the chunk pipeline running across tiles (2x slower), 32-channel chunks, 64-pixel
// tiles at 7 waves/SIMD (10 % slower: W is staged twice as often), staggered workgroup starts (no effect), and X
// straight from global memory in MFMA operand layout with only W in LDS (10-20 % slower).  What did pay: fewer and
// wider vector-memory instructions covering fewer cache lines each (W as one 16-byte load per thread, whole-row
// stores) - the address unit (TA) was the busiest block of the CU at 63 %.
// TINY: rows shorter than 4 pixels (scalar guarded path) - a kernel of its own so that its register needs do not set
// the occupancy of the real one.  Also measured and dropped: two chunks in flight per workgroup (same time).
template <int PT, bool TINY, bool ACT = false, bool DG = false, bool BF = false, int VH = 0>
__global__ __launch_bounds__(256, (TINY ? 2 : 4)) void channel_mix_kernel(ChannelMixParams p) {
    __shared__ __attribute__((aligned(16))) float sX[2][CM_KC * (PT + 16)];
    __shared__ float sW[2][CM_KC * CM_WS];
    __shared__ float4 sVH[VH != 0 ? 64 : 1];            // virtual operand: (w[c][0..2], b[c]) per channel
    if constexpr (VH != 0) {
        const int nvh = VH == 1 ? p.Ci : p.Co1;
        if ((int)threadIdx.x < nvh) {
            const float* wr = p.vh_w + threadIdx.x * p.vh_ci;
            sVH[threadIdx.x] = make_float4(wr[0], p.vh_ci > 1 ? wr[1] : 0.f, p.vh_ci > 2 ? wr[2] : 0.f, p.vh_b ? p.vh_b[threadIdx.x] : 0.f);
        }
        __syncthreads();
    }
    // XCD-aware tile order: workgroups go round-robin to the 8 XCDs (gridDim.x is a multiple of 8), so XCD k gets
    // the k-th contiguous eighth of the tile list.  Neighbouring pixel tiles share the 128-byte lines at their
    // boundary in every row (rows are only 4-byte aligned); on the same XCD they meet in one L2.
    const int bx_ = sweep_x(p.rev);
    const int tile = (bx_ & 7) * p.per_xcd + (bx_ >> 3);
    if (tile >= p.ntile) return;
    const int p0 = (tile / p.ncot) * PT, o0 = (tile % p.ncot) * CM_MT, b = sweep_y(p.rev);
    if constexpr (TINY) {
        channel_mix_tile<0, PT, ACT, DG, BF>(p, sX, sW, p0, o0, b);
    } else {
        // interior path: whole pixel tile, whole 16-channel chunks, and the output channels of the tile end on a wave boundary
        // (a last tile with 16 / 32 / 48 valid channels - the 32-channel input gradient of the lift - keeps its whole waves and
        // idles the others instead of falling back to the guarded path)
        if (p0 + PT <= p.P && (p.Ci & (CM_KC - 1)) == 0 && (o0 + CM_MT <= p.Co || ((p.Co - o0) & 15) == 0))
            channel_mix_tile<2, PT, ACT, DG, BF, VH>(p, sX, sW, p0, o0, b, sVH);
        else channel_mix_tile<1, PT, ACT, DG, BF, VH>(p, sX, sW, p0, o0, b, sVH);
    }
}


// Wide variant for Co % 128 == 0: one workgroup computes 128 output channels of its 128 pixels (each wave two groups of
// 16 channels), so X is staged once instead of twice and every X fragment read from LDS feeds two MFMAs.  Interior
// pixel tiles only; the last (partial) pixel tile of a row runs the guarded 64-channel path twice.
constexpr int CMW_WS = 128 + 16;

// GEN (round 6): the same tile on a pixel WINDOW (ChannelMixParams::pm), with the two 64-channel halves of the tile going to two
// destinations (Co1 = 64 mod 128) and the gelu' epilogue on the first destination - the two input gradients of fc1 from ONE staging of
// the output gradient (the generic kernel ran the 128 outputs as two 64-channel tiles, each staging grad_y again: 808 us for 2.9 GB).
//
// PB (with GEN): the X operand is the projected-back gradient of ChannelMixParams::pb_w2 / pb_g (gelu' at staging time - the loaded
// registers wait for the current chunk's MFMAs anyway -, pb_w2 on the weight rows, pb_g on the finished tile).  The last, partial
// pixel tile runs this path too (pixel quads past the end re-read the last quad and are not stored; P % 4 == 0), not the generic tile.
template <bool BF, bool GEN = false, bool PB = false>
__global__ __launch_bounds__(256, (GEN ? 3 : 4)) void channel_mix_wide_kernel(ChannelMixParams p) {
    using T = typename IoElem<BF>::type;
    constexpr int PT = CM_PT, XS = PT + 16, NM = PT / 16;
    __shared__ __attribute__((aligned(16))) float sX[2][CM_KC * XS];
    __shared__ __attribute__((aligned(16))) float sW[2][CM_KC * CMW_WS];
    const int bx_ = sweep_x(p.rev);
    const int tile = (bx_ & 7) * p.per_xcd + (bx_ >> 3);
    if (tile >= p.ntile) return;
    const int p0 = (tile / p.ncot) * PT, o0 = (tile % p.ncot) * 128, b = sweep_y(p.rev);
    if (!PB && (p0 + PT > p.P || (p.Ci & (CM_KC - 1)) != 0)) {
        auto sWn = reinterpret_cast<float (*)[CM_KC * CM_WS]>(&sW[0][0]);
        if constexpr (GEN) {
            if (p.dgelu_of) channel_mix_tile<1, PT, false, true, BF>(p, sX, sWn, p0, o0, b); else channel_mix_tile<1, PT, false, false, BF>(p, sX, sWn, p0, o0, b);
            __syncthreads();
            if (p.dgelu_of) channel_mix_tile<1, PT, false, true, BF>(p, sX, sWn, p0, o0 + 64, b); else channel_mix_tile<1, PT, false, false, BF>(p, sX, sWn, p0, o0 + 64, b);
            return;
        }
        channel_mix_tile<1, PT, false, false, BF>(p, sX, sWn, p0, o0, b);
        __syncthreads();
        channel_mix_tile<1, PT, false, false, BF>(p, sX, sWn, p0, o0 + 64, b);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C1 = p.C1;
    const int PS = GEN ? p.pm.PS : p.P;                 // elements between two channel planes
    const PixRun run = GEN ? pix_run(p.pm, p0) : PixRun{0, 0x7fffffff, 0};
    const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * C1 * PS;
    const T* xb2 = p.x2 ? reinterpret_cast<const T*>(p.x2) + (size_t)b * (p.Ci - C1) * PS : xb;
    const bool tr = p.w_so == 1 && p.w_si != 1;

    float4 rx[2], rw[2];
    float rs = 1.f;                                     // PB: pb_w2 of this thread's weight row
    // PB: a pixel quad past the end of the (last) tile stands for the last quad of the call
    const int xq = PB ? run(min(p0 + (tid & 31) * 4, p.P - 4)) : 0;
    auto load_chunk = [&](int k0) {
        const T* cb = k0 < C1 ? xb : xb2;
        const int kb = k0 < C1 ? k0 : k0 - C1;
        if constexpr (PB) rs = p.pb_w2[k0 + (tid >> 4)];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u;
            if constexpr (PB) rx[u] = io_ld4(cb + (size_t)(kb + (e >> 5)) * PS + xq);
            else if constexpr (GEN) rx[u] = io_ld4(cb + (size_t)(kb + (e >> 5)) * PS + run(p0 + (e & 31) * 4));
            else rx[u] = io_ld4(cb + (unsigned)((kb + (e >> 5)) * p.P + p0 + (e & 31) * 4));
            const unsigned woff = tr ? (unsigned)((k0 + (tid >> 4)) * p.w_si + o0 + 64 * u + (tid & 15) * 4)
                                     : (unsigned)((o0 + 64 * u + (tid >> 2)) * p.w_so + k0 + (tid & 3) * 4);
            const f4u wv = *reinterpret_cast<const f4u*>(p.w + woff);
            rw[u] = make_float4(wv.v[0], wv.v[1], wv.v[2], wv.v[3]);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u;
            if constexpr (PB) {
                rx[u] = make_float4(cm_dgelu(rx[u].x), cm_dgelu(rx[u].y), cm_dgelu(rx[u].z), cm_dgelu(rx[u].w));
                rw[u].x *= rs; rw[u].y *= rs; rw[u].z *= rs; rw[u].w *= rs;
            }
            *reinterpret_cast<float4*>(&sX[buf][(e >> 5) * XS + (e & 31) * 4]) = rx[u];
            float* d = sW[buf] + (tr ? (tid >> 4) * CMW_WS + 64 * u + (tid & 15) * 4 : (tid & 3) * 4 * CMW_WS + 64 * u + (tid >> 2));
            const int dstep = tr ? 1 : CMW_WS;
            d[0] = rw[u].x; d[dstep] = rw[u].y; d[2 * dstep] = rw[u].z; d[3 * dstep] = rw[u].w;
        }
    };

    f32x4 acc[2][NM];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int mt = 0; mt < NM; ++mt) acc[g][mt] = f32x4{0, 0, 0, 0};

    const int nchunk = p.Ci / CM_KC;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) load_chunk((c + 1) * CM_KC);
#pragma unroll
        for (int ks = 0; ks < CM_KC / 4; ++ks) {
            const float* wrow = sW[buf] + (4 * ks + kk) * CMW_WS + 16 * wave + r16;
            const float w0 = wrow[0], w1 = wrow[64];
            const float* xrow = sX[buf] + (4 * ks + kk) * XS + r16;
#pragma unroll
            for (int mt = 0; mt < NM; ++mt) {
                const float xv = xrow[16 * mt];
                acc[0][mt] = mfma16(xv, w0, acc[0][mt]);
                acc[1][mt] = mfma16(xv, w1, acc[1][mt]);
            }
        }
        if (c + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // epilogue as in the 64-channel kernel, once per channel group
    constexpr int OS = PT + 16;
    float* sO = &sX[0][0] + wave * (8 * OS);
    const int c4 = (lane & 31) * 4;
    const CmDest<T> dd0 = cm_dest<T>(p, o0, b);         // a 128-channel tile lies in one destination (Co1 % 128 == 0) - GEN: one per half
    T* const aall = (!GEN && p.y_act) ? reinterpret_cast<T*>(p.y_act) + (size_t)b * p.Co * p.P : nullptr;
    const bool live = !PB || p0 + c4 < p.P;
    const int poff = PB ? run(min(p0 + c4, p.P - 4)) : GEN ? run(p0 + c4) : p0 + c4;      // this lane's four pixels inside a plane
    float4 gq = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (PB) gq = io_ld4(p.pb_g + (size_t)b * PS + poff);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int ob = o0 + 64 * g + 16 * wave;
        const float bias_l = p.bias ? p.bias[ob + r16] : 0.f;
        const CmDest<T> dd = GEN ? cm_dest<T>(p, o0 + 64 * g, b) : dd0;
        const T* const dall = (GEN && p.dgelu_of && o0 + 64 * g < p.Co1) ? reinterpret_cast<const T*>(p.dgelu_of) + (size_t)b * p.Co1 * PS : nullptr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((r16 >> 3) == h) {
#pragma unroll
                for (int mt = 0; mt < NM; ++mt)
                    *reinterpret_cast<float4*>(sO + (r16 & 7) * OS + 16 * mt + 4 * kk) =
                        make_float4(acc[g][mt][0], acc[g][mt][1], acc[g][mt][2], acc[g][mt][3]);
            }
            __syncthreads();
            float4 old[4], dgv[4];
            T* dst[4];
            size_t aoff[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int o = ob + 8 * h + 2 * it + (lane >> 5);
                dst[it] = dd.base + (size_t)(o - dd.ob) * PS + poff;
                aoff[it] = (size_t)o * p.P + p0 + c4;
                if (p.accumulate) old[it] = io_ld4(dst[it]);
                else old[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (GEN) { if (dall) dgv[it] = io_ld4(dall + (size_t)o * PS + poff); }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = 2 * it + (lane >> 5);
                const float4 v = *reinterpret_cast<const float4*>(sO + row * OS + c4);
                const float bv = __shfl(bias_l, 8 * h + row);
                float t0 = v.x + bv, t1 = v.y + bv, t2 = v.z + bv, t3 = v.w + bv;
                if constexpr (PB) { t0 = fmaf(v.x, gq.x, bv); t1 = fmaf(v.y, gq.y, bv); t2 = fmaf(v.z, gq.z, bv); t3 = fmaf(v.w, gq.w, bv); }
                if constexpr (GEN) {
                    // as the generic kernel's epilogue: the PRODUCT is multiplied by gelu'(dgelu_of), then added to the old value
                    if (dall) { t0 *= cm_dgelu(dgv[it].x); t1 *= cm_dgelu(dgv[it].y); t2 *= cm_dgelu(dgv[it].z); t3 *= cm_dgelu(dgv[it].w); }
                }
                const float w0 = old[it].x + t0, w1 = old[it].y + t1, w2 = old[it].z + t2, w3 = old[it].w + t3;
                if (live) io_store4(dst[it], w0, w1, w2, w3);
                if (aall) io_store4(aall + aoff[it], cm_gelu(w0), cm_gelu(w1), cm_gelu(w2), cm_gelu(w3));
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ K8-S
// Wide layers (Ci >= 128, Co % 128 == 0) on the bf16 matrix pipe with BOTH operands split into three bfloat16 pieces
// (x = hi + mid + lo exactly to 2^-24 |x|; six products hi hi, hi mid, mid hi, hi lo, mid mid, lo hi, f32 accumulation): relative
// error ~1e-7 - measured below the f32 MFMA form's own 3e-7 (profiles/r04_split_bf16_error.txt) - at 6 x 16 cycles per
// 16 x 16 x 32 block instead of 8 x 32.  Why here and not in the transforms: these layers are MFMA-bound, not memory-bound - a
// 256 -> 256 layer at 111^2 is 25.8 GFLOP = 164 us of f32 MFMA peak for 808 MB = 135 us of HBM time, measured 286 us; the 256-channel
// layers of the Darcy step ran at 55-60 % of the f32 MFMA peak.  bf16 activations (BF) are exact in one
// piece: three products.
//   * workgroup = 128 pixels x 128 output channels, K chunks of 32 input channels; wave (a, b) owns pixels 64 a .. + 63 x channels
//     64 b .. + 63 = 4 x 4 accumulator tiles: every operand fragment feeds four MFMA groups;
//   * both operands are split ONCE per workgroup, by the thread that stages them (88 VALU per 16 values), and live in LDS as bf16:
//     - X as three planes [k][px] (the layout it arrives in: a thread's four pixels of one channel are one 8-byte write per
//       plane); the A operand - 8 k-values of one pixel per lane - comes out of two ds_read_b64_tr_b16 (gfx950's transposing read:
//       within a 16-lane group lane 4 j + q supplies the address of segment q of row j and lane i receives column i of the 4 x 16
//       block, tools/probes/ds_read_tr_probe.hip), which numbers a lane's k-slots e = 0..7 as k = 4 g + e | 16 + 4 g + (e - 4);
//       rows 288 bytes apart: the 8 rows x 4 segments of a half-wave cover the 64 banks once;
//     - W as three planes of 16-byte atoms [k-group g][o] holding the SAME k-slots (a thread's four consecutive k of one output
//       channel are half an atom): a lane's B operand is one conflict-free ds_read_b128 per plane and channel tile;
//     no VALU work and 36 LDS reads per 96 MFMAs in the multiply phase.
// One LDS buffer (52 KB), two barriers per chunk, the next chunk's global loads in flight during the MFMAs; two workgroups per CU.
// Non-finite inputs: x - hi(x) is NaN for x = +-inf (the f32 form returns inf where no 0 * inf occurs).
typedef __bf16 cms_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cms_u32x4 __attribute__((ext_vector_type(4)));
typedef short cms_v4i16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 cms_mfma(const cms_u32x4& a, const cms_u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cms_bf16x8, a), __builtin_bit_cast(cms_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void cms_split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_pack2(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = bf16_pack2(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = bf16_pack2(sa, sb);
}
__device__ __forceinline__ uint2 cms_tr_read(const char* lds_addr) {
    typedef cms_v4i16 __attribute__((address_space(3))) * lds_v4;
    const cms_v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(const void __attribute__((address_space(3)))*)lds_addr);
    return __builtin_bit_cast(uint2, v);
}
constexpr int CMS_KC = 32;
constexpr int CMS_XRS = CM_PT * 2 + 32;                 // bytes between two k rows of an X plane
constexpr int CMS_XPLANE = CMS_KC * CMS_XRS;            // 9 216
constexpr int CMS_WPLANE = 4 * 128 * 16;                // bytes per W plane: [k-group 4][o 128] atoms of 8 bf16
constexpr int CMS_FALLBACK = 2 * CM_KC * (CM_PT + 16) * 4 + 2 * CM_KC * CM_WS * 4;      // staging buffers of the guarded fallback path

// the weights of one call as K8-S's W operand: per (channel tile of CT, chunk of 32 input channels) the three bf16 planes
// [k-group g 0..3][o 0..CT-1] of 16-byte atoms holding k = 4 g .. 4 g + 3 (slots 0..3) and 16 + 4 g .. + 3 (slots 4..7) of the chunk -
// the same arithmetic (cms_split3 on the same pairs) and the same bytes the staging threads of the WM = 0 / 1 forms produce.
__global__ __launch_bounds__(256) void channel_mix_wsplit_kernel(const float* __restrict__ w, int w_so, int w_si, int Ci, int Co, int CT,
                                                               cms_u32x4* __restrict__ out) {
    const int nchunk = Ci / CMS_KC;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= (Co / CT) * nchunk * 4 * CT) return;
    const int ol = a % CT, g = (a / CT) & 3, c = (a / (4 * CT)) % nchunk, ot = a / (4 * CT * nchunk);
    const float* wr = w + (size_t)(ot * CT + ol) * w_so;
    const int klo = CMS_KC * c + 4 * g, khi = klo + 16;
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = wr[(size_t)(klo + j) * w_si]; v[4 + j] = wr[(size_t)(khi + j) * w_si]; }
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cms_split3(v[2 * j], v[2 * j + 1], h[j], m[j], l[j]);
    cms_u32x4* img = out + (size_t)(ot * nchunk + c) * (3 * 4 * CT) + g * CT + ol;
    img[0] = cms_u32x4{h[0], h[1], h[2], h[3]};
    img[4 * CT] = cms_u32x4{m[0], m[1], m[2], m[3]};
    img[8 * CT] = cms_u32x4{l[0], l[1], l[2], l[3]};
}

// CT: output channels per tile, 128 or 64 (layers with Co % 64 == 0 only: conv5's 256 -> 64, the input gradients of the 64-channel
// levels).  CT = 64: the four waves are four pixel quarters (32 pixels x 64 channels = 2 x 4 accumulator tiles each).
// ACT: x := gelu(x) for the channels of the first source as they are staged (the layer's input is kept pre-activation: fc1 behind conv5,
// darcy_flow_uno2d.py:126-129); gelu(x) is an f32 value, so bf16 activations take three pieces as well.  Epilogues as in the generic
// kernel: `dgelu_of` (gelu' of the saved pre-activation on the first destination), the fused one-channel projection (CT = 64, Co = 64).
// WM: where the W operand comes from - 0: the (Co, Ci) matrix, 1: its transpose (input-gradient call; both split by the threads that
// stage them), 2 (round 5): a SHADOW of the weights already split into the three bf16 planes and stored as the LDS images of the
// (channel tile, chunk) pairs (channel_mix_wsplit_kernel, one tiny launch per call into caller-provided scratch).  W is the same for
// every pixel tile, yet each of the ~1 400 workgroups per batch entry split it again: 88 of the ~190 VALU instructions a thread spends
// per chunk, in a kernel whose MFMA and VALU time add up.  With the shadow a chunk of W is six (CT = 64: three)
// 16-byte copies per thread and no arithmetic.
template <bool BF, int WM, int XP = 0, int CT = 128, bool ACT = false>      // XP: development knock-outs / stamps
__global__ __launch_bounds__(256, 2) void channel_mix_split_kernel(ChannelMixParams p) {
    constexpr bool tr = WM == 1;
    constexpr bool WSH = WM == 2;
    constexpr int NWS = 3 * 4 * CT / 256;                // 16-byte atoms of a W image per thread: 6 | 3
    constexpr int MW = CT == 128 ? 4 : 2;               // pixel tiles (of 16) per wave
    using T = typename IoElem<BF>::type;
    constexpr int PT = CM_PT;
    constexpr int NPX = (BF && !ACT) ? 1 : 3;           // pieces of the X operand
    constexpr int LDS_MAIN = NPX * CMS_XPLANE + 3 * CMS_WPLANE;
    __shared__ __attribute__((aligned(16))) char smem[LDS_MAIN > CMS_FALLBACK ? LDS_MAIN : CMS_FALLBACK];
    char* sXb = smem;
    char* sWb = smem + NPX * CMS_XPLANE;
    const int bx_ = sweep_x(p.rev);
    const int tile = (bx_ & 7) * p.per_xcd + (bx_ >> 3);
    if (tile >= p.ntile) return;
    const int p0 = (tile / p.ncot) * PT, o0 = (tile % p.ncot) * CT, b = sweep_y(p.rev);
    if (p0 + PT > p.P) {            // the last, partial pixel tile of a row: the guarded 64-channel path (twice for a 128-channel tile, as the wide kernel)
        auto sXn = reinterpret_cast<float (*)[CM_KC * (PT + 16)]>(smem);
        auto sWn = reinterpret_cast<float (*)[CM_KC * CM_WS]>(smem + 2 * CM_KC * (PT + 16) * 4);
        // (the generic tile code with the same flags: ACT as a template parameter, gelu' / projection by its run-time pointers)
        if (p.dgelu_of) {
            channel_mix_tile<1, PT, false, true, BF>(p, sXn, sWn, p0, o0, b);
            if constexpr (CT == 128) { __syncthreads(); channel_mix_tile<1, PT, false, true, BF>(p, sXn, sWn, p0, o0 + 64, b); }
        } else {
            channel_mix_tile<1, PT, ACT, false, BF>(p, sXn, sWn, p0, o0, b);
            if constexpr (CT == 128) { __syncthreads(); channel_mix_tile<1, PT, ACT, false, BF>(p, sXn, sWn, p0, o0 + 64, b); }
        }
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wb = wave >> 1;
    const int wpx = CT == 128 ? 64 * wa : 32 * wave, wch = CT == 128 ? 64 * wb : 0;      // this wave's first pixel / channel inside the tile
    const int C1 = p.C1;
    const int PS = p.pm.PS;
    const PixRun run = pix_run(p.pm, p0);
    const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * C1 * PS;
    const T* xb2 = p.x2 ? reinterpret_cast<const T*>(p.x2) + (size_t)b * (p.Ci - C1) * PS : xb;
    unsigned xoff[4];                                    // a thread's four staged pieces: channel row e >> 5 of the chunk, pixels 4 (e & 31) ..
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = tid + 256 * u; xoff[u] = (unsigned)((e >> 5) * PS + run(p0 + (e & 31) * 4)); }

    float4 rx[4], rw[WSH ? 1 : 4];
    cms_u32x4 rws[WSH ? NWS : 1];
    const cms_u32x4* const wimg = reinterpret_cast<const cms_u32x4*>(p.wsplit) + (size_t)(o0 / CT) * (p.Ci / CMS_KC) * (3 * 4 * CT);
    bool act_ld = false;                                 // (uniform) the chunk in the registers belongs to the first source
    // WM = 2: TWO chunks of X in flight (register sets rx / rx2).  With one, a chunk's loads had the 96 MFMAs of the previous chunk -
    // 0.64 us - to arrive in, less than the memory latency under load: the wave waited at every store (the same finding as K9-S, round 4).
    // The registers come from the W operand, which no longer passes through 16 f32 registers and a split.
    float4 rx2[WSH ? 4 : 1];
    bool act_ld2 = false;
    auto load_x = [&](int k0, float4* r, bool& act) {
        act = ACT && k0 < C1;
        const T* cb = k0 < C1 ? xb : xb2;
        const int kb = k0 < C1 ? k0 : k0 - C1;
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = io_ld4(cb + ((unsigned)(kb * PS) + xoff[u]));
    };
    auto load_w = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NWS; ++u) rws[u] = wimg[(size_t)(k0 / CMS_KC) * (3 * 4 * CT) + tid + 256 * u];
    };
    auto load_chunk = [&](int k0) {
        act_ld = ACT && k0 < C1;
        const T* cb = k0 < C1 ? xb : xb2;                // a 32-channel chunk lies in one source (C1 % 32 == 0)
        const int kb = k0 < C1 ? k0 : k0 - C1;
        if constexpr (WSH) {
#pragma unroll
            for (int u = 0; u < NWS; ++u) rws[u] = wimg[(size_t)(k0 / CMS_KC) * (3 * 4 * CT) + tid + 256 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            if constexpr (!(XP & 4)) rx[u] = io_ld4(cb + ((unsigned)(kb * PS) + xoff[u]));
            else rx[u] = make_float4(1.f, 2.f, 3.f, 4.f);
            if constexpr (WSH) continue;
            if constexpr ((XP & 2) != 0) { rw[u] = make_float4(1.f, 2.f, 3.f, 4.f); continue; }
            if (CT == 64 && u >= 2) continue;             // 64 channels x 32 k = 512 float4: the first two rounds
            if constexpr (CT == 64) {
                if constexpr (!tr) {
                    const int o = (e & 15) | ((e >> 7) << 4), k4 = ((e >> 4) & 7) * 4;        // e < 512: o < 64
                    const f4u wv = *reinterpret_cast<const f4u*>(p.w + (unsigned)((o0 + o) * p.w_so + k0 + k4));
                    rw[u] = make_float4(wv.v[0], wv.v[1], wv.v[2], wv.v[3]);
                } else {
                    const int o = e & 63, k4 = (e >> 6) * 4;
                    const float* wp = p.w + (unsigned)((k0 + k4) * p.w_si + o0 + o);
                    rw[u] = make_float4(wp[0], wp[p.w_si], wp[2 * p.w_si], wp[3 * p.w_si]);
                }
            } else if constexpr (!tr) {
                // 16 consecutive lanes = 16 consecutive output channels at one k-quad (their LDS writes: 2-way, not 4-way), the four
                // lane groups of a wave = 64 contiguous bytes of each of those rows
                const int o = (e & 15) | ((e >> 7) << 4), k4 = ((e >> 4) & 7) * 4;
                const f4u wv = *reinterpret_cast<const f4u*>(p.w + (unsigned)((o0 + o) * p.w_so + k0 + k4));
                rw[u] = make_float4(wv.v[0], wv.v[1], wv.v[2], wv.v[3]);
            } else {
                const int o = e & 127, k4 = (e >> 7) * 4;
                const float* wp = p.w + (unsigned)((k0 + k4) * p.w_si + o0 + o);
                rw[u] = make_float4(wp[0], wp[p.w_si], wp[2 * p.w_si], wp[3 * p.w_si]);
            }
        }
    };
    auto store_x = [&](float4* r, bool act) {           // WM = 2: the X part of store_chunk on a given register set
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            char* d = sXb + (e >> 5) * CMS_XRS + (e & 31) * 8;
            float4 v = r[u];
            if constexpr (ACT) { if (act) v = cm_gelu4(v); }
            if constexpr (NPX == 1) {
                *reinterpret_cast<uint2*>(d) = make_uint2(bf16_pack2(v.x, v.y), bf16_pack2(v.z, v.w));
            } else {
                unsigned h0, m0, l0, h1, m1, l1;
                cms_split3(v.x, v.y, h0, m0, l0);
                cms_split3(v.z, v.w, h1, m1, l1);
                *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(d + (NPX > 1 ? 1 : 0) * CMS_XPLANE) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(d + (NPX > 2 ? 2 : 0) * CMS_XPLANE) = make_uint2(l0, l1);
            }
        }
    };
    auto store_w = [&]() {
        // the image of this (channel tile, chunk): [plane 3][k-group 4][o CT] atoms - for CT = 128 the LDS layout itself
#pragma unroll
        for (int u = 0; u < NWS; ++u) {
            const int e = tid + 256 * u;
            const int off = CT == 128 ? 16 * e : (e >> 8) * CMS_WPLANE + ((((e & 255) >> 6) * 128) + (e & 63)) * 16;
            *reinterpret_cast<cms_u32x4*>(sWb + off) = rws[u];
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            {
                char* d = sXb + (e >> 5) * CMS_XRS + (e & 31) * 8;
                if constexpr (ACT) { if (act_ld) rx[u] = cm_gelu4(rx[u]); }
                if constexpr (NPX == 1) {
                    *reinterpret_cast<uint2*>(d) = make_uint2(bf16_pack2(rx[u].x, rx[u].y), bf16_pack2(rx[u].z, rx[u].w));      // exact: widened bf16
                } else {
                    unsigned h0, m0, l0, h1, m1, l1;
                    if constexpr ((XP & 8) != 0) { h0 = __float_as_uint(rx[u].x); m0 = __float_as_uint(rx[u].y); l0 = h0 ^ m0; h1 = __float_as_uint(rx[u].z); m1 = __float_as_uint(rx[u].w); l1 = h1 ^ m1; }
                    else { cms_split3(rx[u].x, rx[u].y, h0, m0, l0); cms_split3(rx[u].z, rx[u].w, h1, m1, l1); }
                    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(d + (NPX > 1 ? 1 : 0) * CMS_XPLANE) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2*>(d + (NPX > 2 ? 2 : 0) * CMS_XPLANE) = make_uint2(l0, l1);
                }
            }
            if constexpr ((XP & 16) != 0 || WSH) continue;
            if (CT == 64 && u >= 2) continue;
            const int o = tr ? (CT == 64 ? (e & 63) : (e & 127)) : ((e & 15) | ((e >> 7) << 4));
            const int k4 = tr ? (CT == 64 ? (e >> 6) * 4 : (e >> 7) * 4) : ((e >> 4) & 7) * 4;
            unsigned h0, m0, l0, h1, m1, l1;
            cms_split3(rw[u].x, rw[u].y, h0, m0, l0);
            cms_split3(rw[u].z, rw[u].w, h1, m1, l1);
            // k-slots of an atom as the transposing read numbers them: k4 .. k4 + 3 are slots 0..3 (k4 < 16) or 4..7 of k-group (k4 & 15) / 4
            char* d = sWb + (((((k4 & 15) >> 2) * 128) + o) * 16 + (k4 >> 4) * 8);
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + CMS_WPLANE) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(d + 2 * CMS_WPLANE) = make_uint2(l0, l1);
        }
        if constexpr (WSH) store_w();
    };

    f32x4 acc[MW][4];                   // [pixel tile m][channel tile t]
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0, 0, 0, 0};

    // transposing read: lane (group kk, index r16) supplies the address of row 4 kk + (r16 >> 2), segment r16 & 3
    const char* xat = sXb + (4 * kk + (r16 >> 2)) * CMS_XRS + (wpx + 4 * (r16 & 3)) * 2;
    const char* wat = sWb + ((kk * 128 + wch + r16) * 16);
    auto compute = [&]() {
        cms_u32x4 Wop[4][3];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) Wop[t][pl] = *reinterpret_cast<const cms_u32x4*>(wat + pl * CMS_WPLANE + t * 256);
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            cms_u32x4 Xp[NPX];
#pragma unroll
            for (int pl = 0; pl < NPX; ++pl) {
                const uint2 a0 = cms_tr_read(xat + pl * CMS_XPLANE + 32 * m);
                const uint2 a1 = cms_tr_read(xat + pl * CMS_XPLANE + 32 * m + 16 * CMS_XRS);
                Xp[pl] = cms_u32x4{a0.x, a0.y, a1.x, a1.y};
            }
            if constexpr ((XP & 1) != 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t][0] += __uint_as_float(Xp[0][0] ^ Xp[NPX - 1][3] ^ Wop[t][0][1] ^ Wop[t][1][2] ^ Wop[t][2][3]);
                continue;
            }
            // smallest products first; the four channel tiles between two uses of an accumulator
            if constexpr (NPX == 3) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = cms_mfma(Xp[NPX > 2 ? 2 : 0], Wop[t][0], acc[m][t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = cms_mfma(Xp[NPX > 1 ? 1 : 0], Wop[t][1], acc[m][t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = cms_mfma(Xp[0], Wop[t][2], acc[m][t]);
            if constexpr (NPX == 3) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = cms_mfma(Xp[NPX > 1 ? 1 : 0], Wop[t][0], acc[m][t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = cms_mfma(Xp[0], Wop[t][1], acc[m][t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = cms_mfma(Xp[0], Wop[t][0], acc[m][t]);
        }
    };

    const int nchunk = p.Ci / CMS_KC;
    // development (XP & 64): cycle stamps per phase of wave 0 .. 3, summed over the chunks, to the buffer proj_out points at
    constexpr bool stamps = (XP & 64) != 0;
    unsigned long long tS = 0, tB1 = 0, tC = 0, tB2 = 0, t_prev = __builtin_readcyclecounter();
    const unsigned long long t_begin = t_prev;
#define CMS_STAMP(acc_) do { if constexpr (stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); (acc_) += t_ - t_prev; t_prev = t_; } } while (0)
    if constexpr (WSH) {
        auto half_step = [&](int c, float4* r, bool& act) {      // chunk c sits in register set r; r is refilled with chunk c + 2
            store_x(r, act);
            store_w();
            CMS_STAMP(tS);
            __syncthreads();
            CMS_STAMP(tB1);
            if (c + 1 < nchunk) load_w((c + 1) * CMS_KC);
            if (c + 2 < nchunk) load_x((c + 2) * CMS_KC, r, act);
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __builtin_amdgcn_sched_barrier(0);
            CMS_STAMP(tC);
            __syncthreads();
            CMS_STAMP(tB2);
        };
        load_x(0, rx, act_ld);
        load_w(0);
        if (nchunk > 1) load_x(CMS_KC, rx2, act_ld2);
        for (int c = 0; c < nchunk; c += 2) {
            half_step(c, rx, act_ld);
            if (c + 1 < nchunk) half_step(c + 1, rx2, act_ld2);
        }
    } else {
    load_chunk(0);
    for (int c = 0; c < nchunk; ++c) {
        store_chunk();                  // (waits for the chunk's loads)
        CMS_STAMP(tS);
        __syncthreads();
        CMS_STAMP(tB1);
        if (c + 1 < nchunk) load_chunk((c + 1) * CMS_KC);       // global -> registers while this chunk is multiplied
        __builtin_amdgcn_sched_barrier(0);
        compute();
        __builtin_amdgcn_sched_barrier(0);
        CMS_STAMP(tC);
        __syncthreads();                // every wave is done with the chunk: the buffers may be overwritten
        CMS_STAMP(tB2);
    }
    }
    const unsigned long long t_loop_end = t_prev;

    // epilogue: a wave's 16 channels x 64 pixels leave through LDS as 256-byte row segments (four rows per store instruction);
    // D[px = 16 m + 4 kk + r][o = 16 t + r16].  The staging area is PRIVATE to the wave (16 rows x 80 floats), a wave's LDS instructions execute in order, so no workgroup barrier is needed -
    // only that the compiler keeps the order (wave_barrier); the first version synchronised the workgroup 16 times per tile
    // (10 900 of a workgroup's 63 500 cycles were epilogue)
    constexpr int OS = 64 + 16;
    constexpr int WPX = 16 * MW;                        // pixels per wave: 64 | 32
    constexpr int LPR = WPX / 4;                        // lanes per staged row (four pixels each): 16 | 8
    constexpr int RPI = 64 / LPR;                       // rows per store instruction: 4 | 8
    float* sO = reinterpret_cast<float*>(smem) + wave * (16 * OS);
    const int c4 = (lane & (LPR - 1)) * 4;
    const CmDest<T> dd = cm_dest<T>(p, o0, b);          // a tile lies in one destination (Co1 % CT == 0)
    T* const aall = p.y_act ? reinterpret_cast<T*>(p.y_act) + (size_t)b * p.Co * PS : nullptr;
    float bias_l[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bias_l[t] = p.bias ? p.bias[o0 + wch + 16 * t + r16] : 0.f;
    // gelu' of the saved pre-activation: first destination only (o0 < Co1: the whole tile or nothing)
    const T* const dall = (p.dgelu_of && o0 < p.Co1) ? reinterpret_cast<const T*>(p.dgelu_of) + (size_t)b * p.Co1 * PS : nullptr;
    if constexpr (CT == 64) {
        // fused projection fc2(gelu(y)) with one output channel (Co = 64: the wave holds all channels of its 32 pixels): per lane the
        // sum over its four channel tiles, then over the 16 lanes of a k-group (fixed butterfly order), four consecutive pixels per store
        if (p.proj_w) {
            float pw[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) pw[t] = p.proj_w[o0 + 16 * t + r16];
            const float pb = p.proj_b ? p.proj_b[0] : 0.f;
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                float ps[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) sum = fmaf(pw[t], cm_gelu(acc[m][t][r] + bias_l[t]), sum);
                    sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
                    ps[r] = sum + pb;
                }
                if (r16 == 0) io_store4(reinterpret_cast<T*>(p.proj_out) + (size_t)b * PS + run(p0 + wpx + 16 * m + 4 * kk), ps[0], ps[1], ps[2], ps[3]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ob = o0 + wch + 16 * t;
        float* sT = sO;
#pragma unroll
        for (int m = 0; m < MW; ++m)
            *reinterpret_cast<float4*>(sT + r16 * OS + 16 * m + 4 * kk) = make_float4(acc[m][t][0], acc[m][t][1], acc[m][t][2], acc[m][t][3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int NIT = 16 / RPI;                   // store instructions per 16-channel tile: 4 | 2
        float4 old[NIT], pre[NIT];
        T* dst[NIT];
        size_t aoff[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int o = ob + RPI * it + lane / LPR;
            dst[it] = dd.base + (size_t)(o - dd.ob) * PS + run(p0 + wpx + c4);
            aoff[it] = (size_t)o * PS + run(p0 + wpx + c4);
            if (p.accumulate) old[it] = io_ld4(dst[it]);
            else old[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dall) pre[it] = io_ld4(dall + aoff[it]);        // (first destination: channel index o, Co1 rows per batch entry)
            else pre[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = RPI * it + lane / LPR;
            const float4 v = *reinterpret_cast<const float4*>(sT + row * OS + c4);
            const float bv = __shfl(bias_l[t], row);
            float w0 = old[it].x + (v.x + bv), w1 = old[it].y + (v.y + bv), w2 = old[it].z + (v.z + bv), w3 = old[it].w + (v.w + bv);
            if (dall) {
                const float d0 = cm_dgelu(pre[it].x), d1 = cm_dgelu(pre[it].y), d2 = cm_dgelu(pre[it].z), d3 = cm_dgelu(pre[it].w);
                if (p.accumulate == 2) { w0 *= d0; w1 *= d1; w2 *= d2; w3 *= d3; }        // gelu' on the completed sum
                else { w0 = fmaf(v.x + bv, d0, old[it].x); w1 = fmaf(v.y + bv, d1, old[it].y); w2 = fmaf(v.z + bv, d2, old[it].z); w3 = fmaf(v.w + bv, d3, old[it].w); }
            }
            io_store4(dst[it], w0, w1, w2, w3);
            if (aall) io_store4(aall + aoff[it], cm_gelu(w0), cm_gelu(w1), cm_gelu(w2), cm_gelu(w3));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (stamps) {
        if (p.proj_out && lane == 0) {
            unsigned long long* o_ = reinterpret_cast<unsigned long long*>(p.proj_out) + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
            o_[0] = tS; o_[1] = tB1; o_[2] = tC; o_[3] = tB2; o_[4] = __builtin_readcyclecounter() - t_loop_end; o_[5] = t_loop_end - t_begin;
            o_[6] = t_begin; o_[7] = __builtin_amdgcn_s_memrealtime();
        }
    }
#undef CMS_STAMP
}

// Few input channels (the lift's first layer: 3 -> 32 at full resolution).  The tiled kernel stages 16-channel chunks - 13 of 16
// rows of every chunk would be zero fill: 148 us against 79 us for the layer's bytes.  Here a thread owns four pixels, keeps its CI
// input values in registers and walks over the output channels with wave-uniform (scalar) weights: pure streaming, 1 KB of every
// output row per wave instruction.
template <int CI, bool BF>
__global__ __launch_bounds__(256) void channel_mix_few_in_kernel(ChannelMixParams p) {
    using T = typename IoElem<BF>::type;
    const int b = blockIdx.y;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int px = (int)(4 * q);
    if (px >= p.P) return;
    const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * CI * p.P;
    T* yb = reinterpret_cast<T*>(p.y) + (size_t)b * p.Co * p.P;
    const bool full = px + 3 < p.P;
    float xv[CI][4];
#pragma unroll
    for (int i = 0; i < CI; ++i) {
        if (full) {
            const float4 v = io_ld4(xb + (size_t)i * p.P + px);
            xv[i][0] = v.x; xv[i][1] = v.y; xv[i][2] = v.z; xv[i][3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[i][e] = px + e < p.P ? io_widen(xb[(size_t)i * p.P + px + e]) : 0.f;
        }
    }
    for (int o = 0; o < p.Co; ++o) {
        const float bv = p.bias ? p.bias[o] : 0.f;
        float r[4] = {bv, bv, bv, bv};
#pragma unroll
        for (int i = 0; i < CI; ++i) {
            const float wv = p.w[o * p.w_so + i * p.w_si];
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaf(wv, xv[i][e], r[e]);
        }
        T* yrow = yb + (size_t)o * p.P + px;
        if (full) {
            io_store4(yrow, r[0], r[1], r[2], r[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (px + e < p.P) io_store1(yrow + e, r[e]);
        }
    }
}

int launch_channel_mix2(const ChannelMixArgs& a, hipStream_t s) {
    const int B = a.B, Ci = a.Ci, Co = a.Co, bf16 = a.bf16;
    const long long P = a.P;
    const bool two_src = a.x2 != nullptr, two_dst = a.y2 != nullptr;
    ChannelMixParams p;
    p.accumulate = a.accumulate == 2 ? 2 : (a.accumulate ? 1 : 0);
    p.wsplit = nullptr;
    if (p.accumulate == 2 && !a.dgelu_of) { set_error("channel_mix: accumulate = 2 (gelu' on the completed sum) needs dgelu_of"); return -2; }
    p.dgelu_of = a.dgelu_of;
    p.x = a.x; p.x2 = a.x2; p.w = a.w; p.bias = a.bias; p.y = a.y; p.y2 = a.y2; p.y_act = a.y_act;
    p.gmul = a.gmul; p.store_y = a.y ? 1 : 0;
    p.vh_x = a.vh_x; p.vh_w = a.vh_w; p.vh_b = a.vh_b; p.vh_ci = a.vh_ci;
    const int vh = a.vh_x ? a.vh_mode : 0;
    if (vh) {
        const int nvh = vh == 1 ? Ci : (a.y2 ? a.Co1 : Co);
        if ((vh != 1 && vh != 2) || !a.vh_w || a.vh_ci < 1 || a.vh_ci > 3 || nvh > 64 || bf16 || a.win.cols || a.x2 || P < CM_PT ||
            (vh == 1 && !a.act_in) || (vh == 2 && (a.dgelu_of || a.act_in))) {
            set_error("channel_mix: a virtual operand has <= 3 real and <= 64 virtual channels, float32, dense, >= %d pixels; as the input it is read through the GELU, as dgelu_of it replaces that argument", CM_PT);
            return -2;
        }
    }
    if (!a.y) p.y = a.y_act;            // (never stored through: address arithmetic only)
    p.proj_w = a.proj_w; p.proj_b = a.proj_b; p.proj_out = a.proj_out;
    if (a.proj_w && (!a.proj_out || Co > CM_MT || two_dst || a.dgelu_of)) {
        set_error("channel_mix: the fused projection needs proj_out, Co <= %d, one destination and no dgelu_of", CM_MT);
        return -2;
    }
    p.B = B; p.Ci = Ci; p.Co = Co; p.P = (int)P;
    const bool windowed = a.win.cols != 0;
    if (const char* why = pix_window_error(a.win, P)) { set_error("channel_mix: %s", why); return -2; }
    p.pm = pix_map(a.win, P);
    const long long PSl = windowed ? a.win.plane : P;            // elements between two channel planes
    // y_act on padded planes: P = H x W dense pixels -> the top-left corner of (act_plane / act_pitch) x act_pitch planes
    const bool act_pad = a.act_cols != 0;
    p.pm_act = PixMap{0, 0, 0, 0u};
    if (act_pad) {
        if ((!a.y_act && !a.gmul) || (a.y_act && a.gmul) || (!a.y && !a.y_act) || a.accumulate || two_dst || a.dgelu_of || a.proj_w || windowed || bf16 || a.act_cols < 260 || a.act_pitch < a.act_cols || P % a.act_cols || P >= (1LL << 24) ||
            a.act_plane < (P / a.act_cols) * (long long)a.act_pitch || (long long)Co * a.act_plane >= (1LL << 31)) {
            set_error("channel_mix: the padded forms take y_act OR gmul, one plain destination, float32, dense operands, 260 <= W <= Wp, H * W < 2^24 pixels");
            return -2;
        }
        p.pm_act = PixMap{(int)a.act_plane, a.act_cols, a.act_pitch - a.act_cols, (unsigned)(((1ULL << 40) + a.act_cols - 1) / (unsigned long long)a.act_cols)};
    }
    p.C1 = two_src ? a.C1 : Ci;
    p.Co1 = two_dst ? a.Co1 : Co;
    // forward: Wm(o, i) = W[o][i] of a (Co, Ci) matrix; transposed: Wm(o, i) = W[i][o] of an (Ci, Co) matrix
    p.w_so = a.transpose_w ? 1 : Ci;
    p.w_si = a.transpose_w ? Co : 1;
    constexpr int PT = CM_PT;
    const int act_in = a.act_in;
    const void* dgelu_of = a.dgelu_of;
    if (act_in && dgelu_of) { set_error("channel_mix: act_in and dgelu_of are exclusive"); return -2; }
    if (vh == 2) p.dgelu_of = a.vh_x;             // (non-null marks the gelu' epilogue; never read)
    const bool wide = !vh && Co % 128 == 0 && P >= PT && !act_in && !dgelu_of && (!two_dst || p.Co1 % 128 == 0) && !windowed && !act_pad;      // (the wide and few-input kernels are dense only)
    // the general wide form (round 6): windows, the gelu' epilogue on the first destination, destinations split at 64 mod 128 channels
    const bool wide_gen = !wide && !vh && !bf16 && Co % 128 == 0 && P >= PT && !act_in && !act_pad && !a.gmul && !a.proj_w && !a.y_act && a.y &&
                          a.accumulate != 2 && (!two_dst || p.Co1 % 64 == 0) && Ci % CM_KC == 0 && Ci < 128;
    // the projected-back operand (ChannelMixParams::pb_*) exists in that kernel only
    p.pb_w2 = a.pb_w2; p.pb_g = a.pb_g;
    const bool pb = a.pb_w2 != nullptr;
    if (pb && (!a.pb_g || !wide_gen || !a.transpose_w || a.bias || two_src || P % 4)) {
        set_error("channel_mix: the projected-back operand goes with a transposed, bias-free float32 call of < 128 input and a multiple of 128 output channels on >= %d pixels (a multiple of 4)", PT);
        return -3;
    }
    // K8-S: the wide layers whose f32 MFMA time exceeds their memory time (from 128 input channels on)
#ifdef UNO_CMS_DEV        // development build only (tools/dev/mkvariant.py): A/B switch, knock-outs, stamp buffer from the environment
    static const bool split_off = getenv("UNO_CM_SPLIT_OFF") != nullptr;
    static const int cms_exp = getenv("UNO_CMS_EXP") ? atoi(getenv("UNO_CMS_EXP")) : 0;
    p.exp = cms_exp;
    if ((cms_exp & 64) && getenv("UNO_CMS_STAMPS")) p.proj_out = reinterpret_cast<void*>((uintptr_t)strtoull(getenv("UNO_CMS_STAMPS"), nullptr, 0));
#else
    constexpr bool split_off = false;
    p.exp = 0;
#endif
    // f32 activations: from 128 input channels on; bf16 activations move half the bytes, there the f32 MFMA is the ceiling from 32 on
    // (profiles/r04_c5_mixed_kernel_stats.csv: the generic forms on bf16 were 4.8 of the mixed C5 step's 16 ms)
    const bool trw_ = a.transpose_w != 0;
    const bool s_common = !split_off && !act_pad && !vh && P >= PT && Ci >= (bf16 ? 32 : 128) && Ci % CMS_KC == 0 && (!two_src || a.C1 % CMS_KC == 0) && !(act_in && trw_);
#ifdef UNO_CMS_DEV
    static const bool force64 = getenv("UNO_CMS_FORCE64") != nullptr;
#else
    constexpr bool force64 = false;
#endif
    const bool split = s_common && !force64 && Co % 128 == 0 && !a.proj_w && (!two_dst || p.Co1 % 128 == 0);
    // the same on 64-channel tiles: layers with Co % 64 == 0 that are not a multiple of 128 wide (conv5's 256 -> 64, fc1 with its fused
    // projection, the input gradients of the 64-channel levels) - f32-MFMA-bound in the generic kernel (256 -> 64 at 223^2: 26 GFLOP =
    // 166 us of f32 MFMA peak for 1.02 GB)
    const bool split64 = s_common && !split && Co % 64 == 0 && (!two_dst || p.Co1 % 64 == 0) && (!a.proj_w || Co == 64);
    if (two_src && (p.C1 < CM_KC || p.C1 >= Ci || p.C1 % CM_KC)) {
        set_error("channel_mix: a two-source call splits the input channels at a multiple of %d inside (0, Ci) (got %d of %d)", CM_KC, p.C1, Ci);
        return -2;
    }
    if (two_dst && (p.Co1 < CM_MT || p.Co1 >= Co || p.Co1 % CM_MT)) {
        set_error("channel_mix: a two-destination call splits the output channels at a multiple of %d inside (0, Co) (got %d of %d)", CM_MT, p.Co1, Co);
        return -2;
    }
    if (a.y_act && (two_dst || dgelu_of)) { set_error("channel_mix: the activated second output goes with a single destination and no dgelu_of"); return -2; }
    const long long npt = (P + PT - 1) / PT, ncot = (split || ((wide || wide_gen) && !split64)) ? Co / 128 : (Co + CM_MT - 1) / CM_MT;
    if ((long long)Ci * PSl >= (1LL << 30) || (long long)Ci * Co >= (1LL << 30) || npt * ncot > 0x7fffffffLL || B > 65535) {
        set_error("channel_mix: tensor too large (Ci * pixels and Ci * Co must stay below 2^30)");
        return -2;
    }
    p.ncot = (int)ncot; p.ntile = (int)(npt * ncot); p.per_xcd = (p.ntile + 7) / 8;
    p.rev = next_sweep_reversed(SWEEP_K8);
    const int accumulate = p.accumulate;
    if (Ci <= 4 && !accumulate && !act_in && !dgelu_of && P >= 1024 && !two_src && !two_dst && !a.y_act && !a.proj_w && !windowed) {
        ProfScope prof("uno::channel_mix_few_in_kernel", (bf16 ? 2.0 : 4.0) * B * (double)P * (Ci + Co) + 4.0 * Ci * Co, s);
        const dim3 grid((unsigned)((P + 1023) / 1024), B);
#define UNO_CMF(C) do { if (bf16) hipLaunchKernelGGL((channel_mix_few_in_kernel<C, true>), grid, dim3(256), 0, s, p); \
                        else hipLaunchKernelGGL((channel_mix_few_in_kernel<C, false>), grid, dim3(256), 0, s, p); } while (0)
        if (Ci == 1) UNO_CMF(1); else if (Ci == 2) UNO_CMF(2); else if (Ci == 3) UNO_CMF(3); else UNO_CMF(4);
#undef UNO_CMF
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("channel_mix launch: %s", hipGetErrorString(e)); return -5; }
        return 0;
    }
    {
        const double dgc = dgelu_of ? p.Co1 : 0;
        ProfScope prof((split || split64) ? "uno::channel_mix_split_kernel" : (wide || wide_gen) ? "uno::channel_mix_wide_kernel" : "uno::channel_mix_kernel",
                       (bf16 ? 2.0 : 4.0) * B * (double)P * ((vh == 1 ? a.vh_ci : Ci) + (vh == 2 ? a.vh_ci : 0) + (a.y ? Co : 0) + (accumulate ? Co : 0) + dgc + (a.y_act ? Co : 0) + (a.gmul ? Co : 0) + (a.proj_w ? 1 : 0) + (pb ? 1 : 0)) + 4.0 * Ci * Co, s);
        if (split64 || split) {
            const dim3 grid((unsigned)(8 * p.per_xcd), B);
            const bool trw = p.w_so == 1 && p.w_si != 1;
            // the weights pre-split into the kernel's LDS images when the caller provided room for them (6 bytes per weight)
            const int ct = split ? 128 : 64;
            const bool shadow = a.ws && a.ws_bytes >= (size_t)6 * Ci * Co && (reinterpret_cast<uintptr_t>(a.ws) & 15) == 0;
            p.wsplit = shadow ? a.ws : nullptr;
            if (shadow)
                hipLaunchKernelGGL(channel_mix_wsplit_kernel, dim3((unsigned)((Ci / 8 * Co + 255) / 256)), dim3(256), 0, s, p.w, p.w_so, p.w_si, Ci, Co, ct,
                                   reinterpret_cast<cms_u32x4*>(a.ws));
#define UNO_CMS(BF_, TR_, XP_, CT_, ACT_) hipLaunchKernelGGL((channel_mix_split_kernel<BF_, TR_, XP_, CT_, ACT_>), grid, dim3(256), 0, s, p)
#define UNO_CMS_T(BF_, CT_) do { if (shadow) { if (act_in) UNO_CMS(BF_, 2, 0, CT_, true); else UNO_CMS(BF_, 2, 0, CT_, false); } \
                                 else if (act_in) UNO_CMS(BF_, 0, 0, CT_, true); else if (trw) UNO_CMS(BF_, 1, 0, CT_, false); else UNO_CMS(BF_, 0, 0, CT_, false); } while (0)
#ifdef UNO_CMS_DEV          // development build: knock-out / stamp instantiations (f32 activations, 128-channel tiles) selected by UNO_CMS_EXP
#define UNO_CMS_X(XP_) case XP_: if (shadow && XP_ == 64) UNO_CMS(false, 2, 64, 128, false); else if (trw) UNO_CMS(false, 1, XP_, 128, false); else UNO_CMS(false, 0, XP_, 128, false); break;
            if (!bf16 && cms_exp && split && !act_in) {
                switch (cms_exp) {
                    UNO_CMS_X(1) UNO_CMS_X(2) UNO_CMS_X(4) UNO_CMS_X(6) UNO_CMS_X(8) UNO_CMS_X(16) UNO_CMS_X(31) UNO_CMS_X(64) UNO_CMS_X(70)
                    default: UNO_CMS_T(false, 128);
                }
            } else
#undef UNO_CMS_X
#endif
            if (split) { if (bf16) UNO_CMS_T(true, 128); else UNO_CMS_T(false, 128); }
            else { if (bf16) UNO_CMS_T(true, 64); else UNO_CMS_T(false, 64); }
#undef UNO_CMS_T
#undef UNO_CMS
        }
        else if (wide && bf16) hipLaunchKernelGGL(channel_mix_wide_kernel<true>, dim3((unsigned)(8 * p.per_xcd), B), dim3(256), 0, s, p);
        else if (wide) hipLaunchKernelGGL(channel_mix_wide_kernel<false>, dim3((unsigned)(8 * p.per_xcd), B), dim3(256), 0, s, p);
        else if (pb) hipLaunchKernelGGL((channel_mix_wide_kernel<false, true, true>), dim3((unsigned)(8 * p.per_xcd), B), dim3(256), 0, s, p);
        else if (wide_gen) hipLaunchKernelGGL((channel_mix_wide_kernel<false, true>), dim3((unsigned)(8 * p.per_xcd), B), dim3(256), 0, s, p);
        else {
            const dim3 grid((unsigned)(8 * p.per_xcd), B);
#define UNO_CM_LAUNCH(T, BF) \
            do { \
                if (act_in) hipLaunchKernelGGL((channel_mix_kernel<CM_PT, T, true, false, BF>), grid, dim3(256), 0, s, p); \
                else if (dgelu_of) hipLaunchKernelGGL((channel_mix_kernel<CM_PT, T, false, true, BF>), grid, dim3(256), 0, s, p); \
                else hipLaunchKernelGGL((channel_mix_kernel<CM_PT, T, false, false, BF>), grid, dim3(256), 0, s, p); \
            } while (0)
            if (vh == 1) hipLaunchKernelGGL((channel_mix_kernel<CM_PT, false, true, false, false, 1>), grid, dim3(256), 0, s, p);
            else if (vh == 2) hipLaunchKernelGGL((channel_mix_kernel<CM_PT, false, false, true, false, 2>), grid, dim3(256), 0, s, p);
            else if (bf16) { if (P >= 4) UNO_CM_LAUNCH(false, true); else UNO_CM_LAUNCH(true, true); }
            else { if (P >= 4) UNO_CM_LAUNCH(false, false); else UNO_CM_LAUNCH(true, false); }
#undef UNO_CM_LAUNCH
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("channel_mix launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

long long channel_mix_ws_bytes(int Ci, int Co, long long P, int bf16) {
    // (the shapes launch_channel_mix2 can give to K8-S; it decides with the full argument list and ignores the scratch otherwise)
    if (P < CM_PT || Ci < (bf16 ? 32 : 128) || Ci % CMS_KC || Co % 64) return 0;
    return 6LL * Ci * Co;
}

int launch_channel_mix(const void* x, const float* w, const float* bias, void* y, int B, int Ci, int Co, long long P,
                       int transpose_w, int accumulate, int act_in, const void* dgelu_of, int bf16, hipStream_t s, void* ws, size_t ws_bytes) {
    ChannelMixArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.B = B; a.Ci = Ci; a.Co = Co; a.P = P; a.C1 = Ci; a.Co1 = Co;
    a.transpose_w = transpose_w; a.accumulate = accumulate; a.act_in = act_in; a.dgelu_of = dgelu_of; a.bf16 = bf16;
    a.ws = ws; a.ws_bytes = ws_bytes;
    return launch_channel_mix2(a, s);
}

// ------------------------------------------------------------------------------------------------ K9
constexpr int CW_T = 64;            // tile of output channels x tile of input channels per workgroup
constexpr int CW_PK = 32;           // pixels per staged chunk
constexpr int CW_S = CW_PK + 2;     // LDS row stride: 2 r16 + kk hits 32 distinct banks per half-wave

struct ChannelWgradParams {
    const void* gy;         // (B, Co, P) f32 | bf16
    const void* x;          // (B, C1, P) f32 | bf16
    const void* x2;         // (B, Ci - C1, P): input channels [C1, Ci) of a two-source layer (vector kernel only), or nullptr
    int C1;                 // == Ci without a second source
    float* part;            // (nsplit, Co, Ci + 1) partial sums; column Ci holds the bias gradient
    int B, Ci, Co, P, nsplit;
    PixMap pm;              // plane stride + pixel window of gy, x and x2 (vector and split kernels; dense: pm.PS == P)
    const float* vh_x; const float* vh_w; const float* vh_b; int vh_ci;     // vector kernel, VHX: x is VIRTUAL (see ChannelMixParams)
    int act_x;              // scalar kernel: x := gelu(x)
    long long span;         // pixels per split (informational)
    int rev;                // alternating sweep direction (vector and split kernels): pixel splits in descending order
    // split kernel, template PB (round 6): gy is the PRE-ACTIVATION of the layer and the gradient at its output is never stored -
    // gy[b][o][q] stands for pb_w2[o] gelu'(gy[b][o][q]) pb_g[b][q] (see ChannelMixParams::pb_w2); the kernel also leaves the partial sums
    // of the projection's own gradients in part2 (nsplit, Co + 1): sum_q gelu(gy[b][o][q]) pb_g[b][q] per channel, sum_q pb_g[b][q] in slot Co
    const float* pb_w2; const float* pb_g; float* part2;
};

template <bool BF>
__global__ __launch_bounds__(256) void channel_wgrad_kernel(ChannelWgradParams p, int npc, int chunks_per_split) {
    using T = typename IoElem<BF>::type;
    __shared__ float sG[2][CW_T * CW_S];
    __shared__ float sXc[2][CW_T * CW_S];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile_i = (p.Ci + CW_T - 1) / CW_T;
    const int o0 = (blockIdx.x / ntile_i) * CW_T, i0 = (blockIdx.x % ntile_i) * CW_T;
    const int split = blockIdx.y;
    const int c_begin = split * chunks_per_split, c_end = min(c_begin + chunks_per_split, p.B * npc);

    // staging: each operand chunk = 64 rows x 32 px of one batch entry -> 8 elements per thread (row e / 32, px e % 32)
    float rg[8], rxv[8];
    auto load_chunk = [&](int idx) {
        const int b = idx / npc, pp0 = (idx - b * npc) * CW_PK;
        const T* gb = reinterpret_cast<const T*>(p.gy) + (size_t)b * p.Co * p.P;
        const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * p.Ci * p.P;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            const int row = e >> 5, px = pp0 + (e & 31);
            rg[u] = (px < p.P && o0 + row < p.Co) ? io_widen(gb[(size_t)(o0 + row) * p.P + px]) : 0.f;
            rxv[u] = (px < p.P && i0 + row < p.Ci) ? io_widen(xb[(size_t)(i0 + row) * p.P + px]) : 0.f;
            if (p.act_x) rxv[u] = cm_gelu(rxv[u]);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            sG[buf][(e >> 5) * CW_S + (e & 31)] = rg[u];
            sXc[buf][(e >> 5) * CW_S + (e & 31)] = rxv[u];
        }
    };

    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0, 0, 0, 0};
    float bsum = 0.f;                                   // bias gradient: threads 0..63 own one output channel each (i-tile 0 only)

    if (c_begin < c_end) { load_chunk(c_begin); store_chunk(0); }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        const bool more = c + 1 < c_end;
        if (more) load_chunk(c + 1);
#pragma unroll
        for (int ks = 0; ks < CW_PK / 4; ++ks) {
            const float a = sG[buf][(16 * wave + r16) * CW_S + 4 * ks + kk];          // A[o][k = px]
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = mfma16(a, sXc[buf][(16 * nt + r16) * CW_S + 4 * ks + kk], acc[nt]);   // B[k = px][i]
        }
        if (i0 == 0 && tid < CW_T) {
            const float* g = sG[buf] + tid * CW_S;
#pragma unroll
            for (int k = 0; k < CW_PK; ++k) bsum += g[k];
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // D[o = 16 wave + 4 kk + r][i = 16 nt + r16]
    float* part = p.part + (size_t)split * p.Co * (p.Ci + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + 16 * wave + 4 * kk + r;
        if (o < p.Co) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int i = i0 + 16 * nt + r16;
                if (i < p.Ci) part[(size_t)o * (p.Ci + 1) + i] = acc[nt][r];
            }
        }
    }
    if (i0 == 0 && tid < CW_T && o0 + tid < p.Co) part[(size_t)(o0 + tid) * (p.Ci + 1) + p.Ci] = bsum;
}

// Vector variant (P >= 64): 64-pixel chunks enumerated per batch entry, 16-byte loads, next chunk in registers
// (32 dwords per thread in flight), bias partial sums taken from the registers on their way to LDS.
constexpr int CWV_PK = 64;
constexpr int CWV_S = CWV_PK + 4;       // 272-byte rows: 16-byte aligned for ds_write_b128; fragment reads hit banks 4 r16 + kk,
                                        // distinct over all 64 lanes (gfx950 LDS: 64 banks; a stride of 66 cost one conflict cycle per read)

// NI: 16-channel tiles of the INPUT side per workgroup, 4 or - layers with at most 32 input channels (the lift's fc0: 32 -> 64 at full
// resolution) - 2: the 64-wide tile spent half of its MFMAs, X loads, GELUs and LDS writes on channels that do not exist.
template <bool ACTX, bool BF, int NI = 4, bool VHX = false>          // ACTX: x := gelu(x) on its way to LDS (the layer's input is kept pre-activation)
__global__ __launch_bounds__(256, 4) void channel_wgrad_vec_kernel(ChannelWgradParams p, int npc, int chunks_per_split) {
    using T = typename IoElem<BF>::type;
    constexpr int ES = BF ? 2 : 4;          // bytes per element
    __shared__ __attribute__((aligned(16))) float sG[CW_T * CWV_S];
    __shared__ __attribute__((aligned(16))) float sXc[CW_T * CWV_S];
    __shared__ float4 sVH[VHX ? 64 : 1];
    if constexpr (VHX) {
        if ((int)threadIdx.x < p.Ci) {
            const float* wr = p.vh_w + threadIdx.x * p.vh_ci;
            sVH[threadIdx.x] = make_float4(wr[0], p.vh_ci > 1 ? wr[1] : 0.f, p.vh_ci > 2 ? wr[2] : 0.f, p.vh_b ? p.vh_b[threadIdx.x] : 0.f);
        }
        __syncthreads();
    }
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile_i = (p.Ci + CW_T - 1) / CW_T;
    const int ntile = ((p.Co + CW_T - 1) / CW_T) * ntile_i;
    // XCD-aware order: workgroups go round-robin to the 8 XCDs; all weight tiles of one pixel split (they read the same
    // gy / x rows) are given to one XCD so that the re-reads hit its L2 (measured before: 1.78x the algorithmic bytes
    // fetched over the fabric for a 2-tile layer)
    const int bxr = sweep_x(p.rev);
    const int xcd = bxr & 7, j = bxr >> 3;
    const int split = (j / ntile) * 8 + xcd, tile = j % ntile;
    if (split >= p.nsplit) return;
    const int o0 = (tile / ntile_i) * CW_T, i0 = (tile % ntile_i) * CW_T;
    const int c_begin = split * chunks_per_split, c_end = min(c_begin + chunks_per_split, p.B * npc);

    // The loads deliver raw 16-byte pieces from clamped addresses; the zero-fill past the row end (shift network) and past
    // the channel count is applied when the chunk goes to LDS, AFTER the MFMA block - applied at load time it consumed the
    // loaded registers at once and the wave waited for its loads (s_waitcnt vmcnt(0)) before every MFMA block.
    float4 rg[4], rxv[4];
    float4 rvx[3];          // VHX: the raw pieces of the real channels at this thread's four pixels
    int sh_cur = 0;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    const int c4 = (tid & 15) * 4, row0 = tid >> 4;
    // buffer loads (uniform base in SGPRs + 32-bit lane offset): the intrinsic is a fixed 128-bit access - as plain loads
    // of a 4-byte-aligned struct the compiler split these into pairs of 8-byte loads once the registers had to stay
    // live across the MFMA block, doubling the vector-memory instructions
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // the 64 input channels of this tile lie in one source (C1 % 64 == 0 in two-source calls): xs = its tensor, Cs its channel
    // count, il the tile's first channel inside it; the GELU-on-read form applies to the first source only
    const bool src2 = i0 >= p.C1;
    const T* xs = reinterpret_cast<const T*>(src2 ? p.x2 : p.x);
    const int Cs = src2 ? p.Ci - p.C1 : p.C1, il = src2 ? i0 - p.C1 : i0;
    const bool actx = ACTX && !src2;
    auto load_chunk = [&](int idx) {
        const int b = idx / npc, pp = (idx - b * npc) * CWV_PK;
        const int PS = p.pm.PS;
        const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const T*>(p.gy) + (size_t)b * p.Co * PS), 0, p.Co * PS * ES, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(xs + (size_t)b * Cs * PS), 0, Cs * PS * ES, 0x00020000);
        const int px = pp + c4, pl = min(px, p.P - 4);
        sh_cur = px - pl;
        const int pc = pix_run(p.pm, pp)(pl);            // offset of the piece inside its channel plane
        if constexpr (VHX) {
            const __amdgpu_buffer_rsrc_t rv_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vh_x + (size_t)b * p.vh_ci * p.P), 0, p.vh_ci * p.P * 4, 0x00020000);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rv_, (min(kx, p.vh_ci - 1) * p.P + pl) * 4, 0, 0);
                rvx[kx] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = row0 + 16 * u;
            if constexpr (BF) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 tg = __builtin_amdgcn_raw_buffer_load_b64(rg_, (min(o0 + row, p.Co - 1) * PS + pc) * 2, 0, 0);
                rg[u] = make_float4(__uint_as_float(tg.x << 16), __uint_as_float(tg.x & 0xffff0000u), __uint_as_float(tg.y << 16), __uint_as_float(tg.y & 0xffff0000u));
                if (u < NI) {
                    const u32x2 tx = __builtin_amdgcn_raw_buffer_load_b64(rx_, (min(il + row, Cs - 1) * PS + pc) * 2, 0, 0);
                    rxv[u] = make_float4(__uint_as_float(tx.x << 16), __uint_as_float(tx.x & 0xffff0000u), __uint_as_float(tx.y << 16), __uint_as_float(tx.y & 0xffff0000u));
                }
            } else {
                const u32x4 tg = __builtin_amdgcn_raw_buffer_load_b128(rg_, (min(o0 + row, p.Co - 1) * PS + pc) * 4, 0, 0);
                rg[u] = make_float4(__uint_as_float(tg.x), __uint_as_float(tg.y), __uint_as_float(tg.z), __uint_as_float(tg.w));
                if (u < NI && !VHX) {
                    const u32x4 tx = __builtin_amdgcn_raw_buffer_load_b128(rx_, (min(il + row, Cs - 1) * PS + pc) * 4, 0, 0);
                    rxv[u] = make_float4(__uint_as_float(tx.x), __uint_as_float(tx.y), __uint_as_float(tx.z), __uint_as_float(tx.w));
                }
            }
        }
    };
    auto shifted = [&](const float4& v, bool valid) {
        float t0 = v.x, t1 = v.y, t2 = v.z, t3 = v.w;
        if (sh_cur & 1) { t0 = t1; t1 = t2; t2 = t3; t3 = 0.f; }
        if (sh_cur & 2) { t0 = t2; t1 = t3; t2 = 0.f; t3 = 0.f; }
        if (sh_cur >= 4 || !valid) { t0 = 0.f; t1 = 0.f; t2 = 0.f; t3 = 0.f; }
        return make_float4(t0, t1, t2, t3);
    };
    auto store_chunk = [&]() {
        float4 q[3];
        if constexpr (VHX) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) { q[kx] = shifted(rvx[kx], kx < p.vh_ci); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = row0 + 16 * u;
            const float4 g = shifted(rg[u], o0 + row < p.Co);
            *reinterpret_cast<float4*>(sG + row * CWV_S + c4) = g;
            bs[u] += (g.x + g.y) + (g.z + g.w);
            if (u < NI) {
                float4 v;
                if constexpr (VHX) {
                    // (pixels past the row end carry the bias instead of zero: they meet the zero fill of gy)
                    const float4 t = sVH[min(il + row, p.Ci - 1)];
                    v = make_float4(fmaf(t.z, q[2].x, fmaf(t.y, q[1].x, fmaf(t.x, q[0].x, t.w))), fmaf(t.z, q[2].y, fmaf(t.y, q[1].y, fmaf(t.x, q[0].y, t.w))),
                                    fmaf(t.z, q[2].z, fmaf(t.y, q[1].z, fmaf(t.x, q[0].z, t.w))), fmaf(t.z, q[2].w, fmaf(t.y, q[1].w, fmaf(t.x, q[0].w, t.w))));
                    if (il + row >= Cs) v = make_float4(0.f, 0.f, 0.f, 0.f);
                } else v = shifted(rxv[u], il + row < Cs);
                if constexpr (ACTX) { if (actx) v = cm_gelu4(v); }             // gelu(0) = 0: the zero fill survives
                *reinterpret_cast<float4*>(sXc + row * CWV_S + c4) = v;
            }
        }
    };

    f32x4 acc[NI];
#pragma unroll
    for (int nt = 0; nt < NI; ++nt) acc[nt] = f32x4{0, 0, 0, 0};

    if (c_begin < c_end) load_chunk(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        store_chunk();
        __syncthreads();
        load_chunk(min(c + 1, c_end - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < CWV_PK / 4; ++ks) {
            const float a = sG[(16 * wave + r16) * CWV_S + 4 * ks + kk];
#pragma unroll
            for (int nt = 0; nt < NI; ++nt)
                acc[nt] = mfma16(a, sXc[(16 * nt + r16) * CWV_S + 4 * ks + kk], acc[nt]);
        }
        __syncthreads();
    }

    float* part = p.part + (size_t)split * p.Co * (p.Ci + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + 16 * wave + 4 * kk + r;
        if (o < p.Co) {
#pragma unroll
            for (int nt = 0; nt < NI; ++nt) {
                const int i = i0 + 16 * nt + r16;
                if (i < p.Ci) part[(size_t)o * (p.Ci + 1) + i] = acc[nt][r];
            }
        }
    }
    if (i0 == 0) {          // bias gradient: the 16 threads that share a row hold its partial sums
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = bs[u];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            const int o = o0 + row0 + 16 * u;
            if ((tid & 15) == 0 && o < p.Co) part[(size_t)o * (p.Ci + 1) + p.Ci] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ K9-S
// The weight gradient of the wide layers (both channel counts >= 96) on the bf16 matrix pipe with both operands split into three
// bfloat16 pieces - the arrangement of K8-S (six products, f32 accumulation, ~1e-7 relative: profiles/r04_split_bf16_error.txt).
// Why: gW = sum over 16 x P pixels of gY X^T has Co x Ci outputs for 4 (Co + Ci) bytes per pixel - 2 Co Ci / (4 (Co + Ci)) flop/B =
// 64 at 256 x 256, 43 at 128 x 256, 21 at 64 x 128 (fc1, + its GELU on read) - against the f32 ridge of 20: the vector kernel above ran
// these layers at 1.5-3.6 TB/s, i.e. at the f32 MFMA peak (256 x 256 at 111^2: 25.8 GFLOP, 258 us; fc1 at 446^2: 52 GFLOP + 204 M
// GELUs, 673 us for 2.44 GB).
//   * workgroup = (32 MR) x 128 weight tile x one split of the pixels, MR = 4 (Co >= 96) or 2 (Co <= 64 .. 95: fc1 128 -> 64, conv5's
//     256 -> 64); wave (a, b) owns output channels 16 MR a .. x input channels 64 b .. + 63 = MR x 4 accumulator tiles; K = pixels,
//     staged 32 at a time;
//   * BOTH operands have their k axis (pixels) contiguous in memory, so the MFMA operand of a lane - 8 consecutive k of one row - is
//     16 contiguous bytes of an LDS row: planes [piece][row: 128 gY + 128 X][32 px] of bf16, rows 80 bytes apart (the 16 rows of a
//     fragment read cover the 64 banks once), one ds_read_b128 per fragment and piece, no transposing reads;
//   * the split happens once per element, in the thread that stages it (as K8-S); the next 32 pixels are in flight in registers
//     while the current ones are multiplied; 60 KB of LDS, two workgroups per CU.
// Partial sums leave in the (split, Co, Ci + 1) layout of the other first-stage kernels; the second stage is shared.
constexpr int CWS_T = 128;
constexpr int CWS_PK = 32;
constexpr int CWS_RS = CWS_PK * 2 + 16;             // bytes per row of a plane
constexpr int CWS_PLANE = 2 * CWS_T * CWS_RS;       // 20 480

// From 100 000 pixels per launch: below, a launch is a handful of 128 x 128 tiles with a few chunks each and the 60 KB workgroups lose to
// the vector kernel (A/B on one box: the NS-2D roll-out - 64^2 .. 16^2 grids at batch 32, at most 74 000 pixels per call - 80.7 ms
// per step with this form on its wide layers, 79.5 without; the Darcy model's smallest level is 197 000).
static bool wgrad_split_shape(int B, int Ci, int Co, long long P) {
#ifdef UNO_CMS_DEV
    static const bool off = getenv("UNO_CW_SPLIT_OFF") != nullptr;         // development build only: A/B against the f32-MFMA form
#else
    constexpr bool off = false;
#endif
    return !off && Ci >= 96 && Co >= 48 && P >= 64 && (long long)B * P >= 100000;
}
static int wgrad_split_rows(int Co) { return Co >= 96 ? CWS_T : 64; }      // output channels per weight tile

// BF: bfloat16 activations (exact in ONE piece: gY x X is one product; with the GELU applied on read, gelu(x) is an f32 value again: three
// pieces of X against the one of gY)
// PB: the gy operand is the projected-back gradient of ChannelWgradParams::pb_* - gelu' (and gelu, for the projection's weight gradient)
// where the staged quad is split, pb_g as a fifth staged row, pb_w2 on the finished sums.
template <bool ACTX, int MR, bool BF, bool PB = false>
__global__ __launch_bounds__(256, 2) void channel_wgrad_split_kernel(ChannelWgradParams p, int npc, int chunks_per_split) {
    static_assert(!(PB && BF), "the projected-back operand is float32");
    __shared__ __attribute__((aligned(16))) char smem[3 * CWS_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wb = wave >> 1;
    const int ntile_i = (p.Ci + CWS_T - 1) / CWS_T;
    constexpr int TO = 32 * MR;                                     // output channels per tile
    using T = typename IoElem<BF>::type;
    constexpr int ES = BF ? 2 : 4;
    constexpr int NPG = BF ? 1 : 3, NPX = (BF && !ACTX) ? 1 : 3;    // bf16 pieces of the two operands
    const int ntile = ((p.Co + TO - 1) / TO) * ntile_i;
    const int bxr = sweep_x(p.rev);
    const int xcd = bxr & 7, j = bxr >> 3;                          // all weight tiles of one pixel split on one XCD (as the vector kernel)
    const int split = (j / ntile) * 8 + xcd, tile = j % ntile;
    if (split >= p.nsplit) return;
    const int o0 = (tile / ntile_i) * TO, i0 = (tile % ntile_i) * CWS_T;
    const int c_begin = split * chunks_per_split, c_end = min(c_begin + chunks_per_split, p.B * npc);

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int c4 = (tid & 7) * 4, row0 = tid >> 3;                  // a thread stages four pixels of rows row0 + 32 u of both operands
    // the 32 input channels of band u lie in one source (C1 % 32 == 0 in two-source calls)
    const T* xsrc[4];
    int xcs[4], xrow[4];
    bool xact[4], xok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ci = i0 + 32 * u;
        const bool s2 = ci >= p.C1;
        xsrc[u] = reinterpret_cast<const T*>(s2 ? p.x2 : p.x);
        xcs[u] = s2 ? p.Ci - p.C1 : p.C1;
        const int il = (s2 ? ci - p.C1 : ci) + row0;
        xok[u] = ci + row0 < p.Ci;
        xrow[u] = min(il, xcs[u] - 1);
        xact[u] = ACTX && !s2;
        if (ci >= p.Ci) { xsrc[u] = reinterpret_cast<const T*>(p.x); xcs[u] = p.C1; xrow[u] = 0; }
    }
    // two half chunks in flight (register sets 0 / 1): with one, the loads had the 96 MFMAs of ONE half (~0.8 us) to arrive in and the
    // wave waited for them at every store (256 x 256 at 111^2: 173 us at 45 % MFMA-pipe use)
    u32x4 rqs[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};  // PB: the four pixels' pb_g
    float bs2[4] = {0.f, 0.f, 0.f, 0.f}, qs = 0.f;                  // PB: sums of gelu(pre) pb_g per staged row, of pb_g
    u32x4 rgs[2][4], rxs[2][4];                                     // RAW loaded pieces: anything computed from them at load time makes the wave wait for its loads at once
    int shs[2] = {0, 0};
    // the zero fill past the row end / past the channel counts is needed only in the last half chunk of a row and in partial weight
    // tiles - both wave-uniform; everywhere else the staged values go straight to the split (12 of 34 VALU instructions per four values)
    bool tails[2] = {false, false};
    const bool edge = o0 + TO > p.Co || i0 + CWS_T > p.Ci;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_half = [&](int it, u32x4 (&rg)[4], u32x4 (&rxv)[4], u32x4& rq, int& sh_cur, bool& tail) {     // half chunk it: 32 pixels of chunk it >> 1
        const int idx = it >> 1;
        const int b = idx / npc, pp = (idx - b * npc) * CWV_PK + (it & 1) * CWS_PK;
        const int PS = p.pm.PS;
        const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const T*>(p.gy) + (size_t)b * p.Co * PS), 0, p.Co * PS * ES, 0x00020000);
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        auto raw2 = [](const u32x2& t) { return u32x4{t.x, t.y, 0u, 0u}; };
        const int px = pp + c4, pl = min(px, p.P - 4);
        sh_cur = px - pl;
        tail = pp + CWS_PK > p.P;
        const int pc = pix_run(p.pm, pp)(pl);            // offset of the piece inside its channel plane
        if constexpr (PB) {
            const __amdgpu_buffer_rsrc_t rq_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.pb_g + (size_t)b * PS), 0, PS * 4, 0x00020000);
            rq = __builtin_amdgcn_raw_buffer_load_b128(rq_, pc * 4, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc[u] + (size_t)b * xcs[u] * PS), 0, xcs[u] * PS * ES, 0x00020000);
            if constexpr (BF) {
                if (u < MR) rg[u] = raw2(__builtin_amdgcn_raw_buffer_load_b64(rg_, (min(o0 + row0 + 32 * u, p.Co - 1) * PS + pc) * 2, 0, 0));
                rxv[u] = raw2(__builtin_amdgcn_raw_buffer_load_b64(rx_, (xrow[u] * PS + pc) * 2, 0, 0));
            } else {
                if (u < MR) rg[u] = __builtin_amdgcn_raw_buffer_load_b128(rg_, (min(o0 + row0 + 32 * u, p.Co - 1) * PS + pc) * 4, 0, 0);
                rxv[u] = __builtin_amdgcn_raw_buffer_load_b128(rx_, (xrow[u] * PS + pc) * 4, 0, 0);
            }
        }
    };
    auto shifted = [&](const u32x4& r, bool valid, int sh_cur, bool slow) {    // slow: zero fill past the row end / past the channel count (see the vector kernel)
        float t0, t1, t2, t3;
        if constexpr (BF) { t0 = __uint_as_float(r.x << 16); t1 = __uint_as_float(r.x & 0xffff0000u); t2 = __uint_as_float(r.y << 16); t3 = __uint_as_float(r.y & 0xffff0000u); }
        else { t0 = __uint_as_float(r.x); t1 = __uint_as_float(r.y); t2 = __uint_as_float(r.z); t3 = __uint_as_float(r.w); }
        if (slow) {
            if (sh_cur & 1) { t0 = t1; t1 = t2; t2 = t3; t3 = 0.f; }
            if (sh_cur & 2) { t0 = t2; t1 = t3; t2 = 0.f; t3 = 0.f; }
            if (sh_cur >= 4 || !valid) { t0 = 0.f; t1 = 0.f; t2 = 0.f; t3 = 0.f; }
        }
        return make_float4(t0, t1, t2, t3);
    };
    auto put1 = [&](char* d, const float4& v) {                    // widened bf16 values: exact in one piece
        *reinterpret_cast<uint2*>(d) = make_uint2(bf16_pack2(v.x, v.y), bf16_pack2(v.z, v.w));
    };
    auto put3 = [&](char* d, const float4& v) {
        unsigned h0, m0, l0, h1, m1, l1;
        cms_split3(v.x, v.y, h0, m0, l0);
        cms_split3(v.z, v.w, h1, m1, l1);
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + CWS_PLANE) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(d + 2 * CWS_PLANE) = make_uint2(l0, l1);
    };
    // Position of a row's four 16-byte k-groups inside its 64 bytes: rows 4 .. 11 of every 16 keep them pairwise swapped.  ds_read_b128 is
    // serviced in lane groups {0-3, 12-15, 20-27}, ... - rows 0-3 / 12-15 at k-group g together with rows 4-11 at k-group g ^ 1 - and with
    // 80-byte rows in plain order three of the 16 accesses of every group met another one's banks (PMC, 256 x 256 at 111^2:
    // SQ_LDS_BANK_CONFLICT 19.0 M of 38.0 M LDS cycles); with the swap the 16 four-bank windows of a group are distinct.
    const int wpos = 16 * ((c4 >> 2 >> 1) ^ (((row0 & 15) + 4) >> 3 & 1)) + 8 * ((c4 >> 2) & 1);       // rows row0 + 32 u: the same row0 & 15
    auto store_half = [&](const u32x4 (&rg)[4], const u32x4 (&rxv)[4], const u32x4& rq, int sh_cur, bool slow) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PB) { q = shifted(rq, true, sh_cur, slow); qs += (q.x + q.y) + (q.z + q.w); }      // (zero past the row end: the products below vanish there)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = row0 + 32 * u;
            if (u < MR) {
                float4 g = shifted(rg[u], o0 + row < p.Co, sh_cur, slow);
                if constexpr (PB) {
                    // (the four products are summed like the bias gradient's below, not fma-chained into bs2[u]: the chained form was packed by
                    // the SLP vectoriser into v_pk_fma_f32 on the (bs2[0], bs2[1]) pair and - in the ACTX = false, MR = 2 instantiation only -
                    // lanes 48-63 of a wave lost one half chunk's terms of bs2[0] in about every fourth launch; -fno-slp-vectorize and this
                    // form are both clean over hundreds of launches: tests/test_hip_project_backward.py repeats the launch)
                    auto back = [&](float v, float qv, float& t) {
                        const float cdf = 0.5f * (1.f + uno_erf(v * 0.70710678118654752440f));
                        const float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
                        t = v * cdf * qv;
                        return fmaf(v, pdf, cdf) * qv;
                    };
                    float t0, t1, t2, t3;
                    g = make_float4(back(g.x, q.x, t0), back(g.y, q.y, t1), back(g.z, q.z, t2), back(g.w, q.w, t3));
                    bs2[u] += (t0 + t1) + (t2 + t3);
                }
                if constexpr (NPG == 1) put1(smem + row * CWS_RS + wpos, g); else put3(smem + row * CWS_RS + wpos, g);
                bs[u] += (g.x + g.y) + (g.z + g.w);
            }
            float4 v = shifted(rxv[u], xok[u], sh_cur, slow);
            if constexpr (ACTX) { if (xact[u]) v = cm_gelu4(v); }   // gelu(0) = 0: the zero fill survives
            if constexpr (NPX == 1) put1(smem + (CWS_T + row) * CWS_RS + wpos, v); else put3(smem + (CWS_T + row) * CWS_RS + wpos, v);
        }
    };

    f32x4 acc[MR][4];                   // [output-channel tile m][input-channel tile t]
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0, 0, 0, 0};
    const int rpos = 16 * (kk ^ ((r16 + 4) >> 3 & 1));
    const char* abase = smem + (16 * MR * wa + r16) * CWS_RS + rpos;
    const char* bbase = smem + (CWS_T + 64 * wb + r16) * CWS_RS + rpos;
    auto compute = [&]() {
        cms_u32x4 A[MR][NPG];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int pl = 0; pl < NPG; ++pl) A[m][pl] = *reinterpret_cast<const cms_u32x4*>(abase + pl * CWS_PLANE + m * 16 * CWS_RS);
        if constexpr (MR >= 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                cms_u32x4 Bp[NPX];
#pragma unroll
                for (int pl = 0; pl < NPX; ++pl) Bp[pl] = *reinterpret_cast<const cms_u32x4*>(bbase + pl * CWS_PLANE + t * 16 * CWS_RS);
                // products (i, j) with i + j <= 2, smallest first; the MR row tiles between two uses of an accumulator
#pragma unroll
                for (int sum = 2; sum >= 0; --sum)
#pragma unroll
                    for (int i = NPG - 1; i >= 0; --i) {
                        const int jx = sum - i;
                        if (jx < 0 || jx >= NPX) continue;
#pragma unroll
                        for (int m = 0; m < MR; ++m) acc[m][t] = cms_mfma(A[m][i], Bp[jx], acc[m][t]);
                    }
            }
        } else {
            // two row tiles: with the column tiles in the outer loop an accumulator came round again after TWO MFMAs (32 cycles, less than
            // the instruction's latency: cycle stamps of the -DUNO_CWS_STAMPS build, fc1 128 -> 64 at 446^2: 2 320 cycles per half chunk
            // for 48 MFMAs = 768 cycles of matrix pipe).  All four column tiles' fragments are read first and the column tiles run
            // inside each product: eight MFMAs between two uses of an accumulator.
            cms_u32x4 Bp[4][NPX];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int pl = 0; pl < NPX; ++pl) Bp[t][pl] = *reinterpret_cast<const cms_u32x4*>(bbase + pl * CWS_PLANE + t * 16 * CWS_RS);
#pragma unroll
            for (int sum = 2; sum >= 0; --sum)
#pragma unroll
                for (int i = NPG - 1; i >= 0; --i) {
                    const int jx = sum - i;
                    if (jx < 0 || jx >= NPX) continue;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int m = 0; m < MR; ++m) acc[m][t] = cms_mfma(A[m][i], Bp[t][jx], acc[m][t]);
                }
        }
    };

    const int it_begin = 2 * c_begin, it_end = 2 * c_end;          // an even number of half chunks
    if (it_begin < it_end) {
        load_half(it_begin, rgs[0], rxs[0], rqs[0], shs[0], tails[0]);
        __builtin_amdgcn_sched_barrier(0);          // set 0's loads strictly before set 1's: the loop's vmcnt waits are derived from BOTH orders
        load_half(it_begin + 1, rgs[1], rxs[1], rqs[1], shs[1], tails[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // development build (-DUNO_CWS_STAMPS, tools/dev/mkvariant.py): cycles per phase of every wave of one workgroup, summed over its
    // half chunks - store (incl. the wait for the loads), barrier, fragment reads + MFMAs, barrier - printed at the end of the kernel
#ifdef UNO_CWS_STAMPS
    unsigned long long tS = 0, tB1 = 0, tC = 0, tB2 = 0, t_prev = __builtin_readcyclecounter();
    const unsigned long long t_begin = t_prev;
#define CWS_STAMP(acc_) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); (acc_) += t_ - t_prev; t_prev = t_; } while (0)
#else
#define CWS_STAMP(acc_) do { } while (0)
#endif
    for (int it = it_begin; it < it_end; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (tails[h] || edge) store_half(rgs[h], rxs[h], rqs[h], shs[h], true);          // (waits for this half chunk's loads only: vmcnt counts the other set's)
            else store_half(rgs[h], rxs[h], rqs[h], shs[h], false);
            CWS_STAMP(tS);
            __syncthreads();
            CWS_STAMP(tB1);
            load_half(min(it + h + 2, it_end - 1), rgs[h], rxs[h], rqs[h], shs[h], tails[h]);
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __builtin_amdgcn_sched_barrier(0);
            CWS_STAMP(tC);
            __syncthreads();
            CWS_STAMP(tB2);
        }
    }
#ifdef UNO_CWS_STAMPS
    if (blockIdx.x == 9 && lane == 0)
        printf("K9-S stamps wg %d wave %d halves %d: store %llu  barrier1 %llu  multiply %llu  barrier2 %llu  loop %llu cycles\n", (int)blockIdx.x, wave,
               it_end - it_begin, tS, tB1, tC, tB2, t_prev - t_begin);
#endif
#undef CWS_STAMP

    float* part = p.part + (size_t)split * p.Co * (p.Ci + 1);
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * MR * wa + 16 * m + 4 * kk + r;
            if (o < p.Co) {
                const float sc = PB ? p.pb_w2[o] : 1.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = i0 + 64 * wb + 16 * t + r16;
                    if (i < p.Ci) part[(size_t)o * (p.Ci + 1) + i] = PB ? acc[m][t][r] * sc : acc[m][t][r];
                }
            }
        }
    if (i0 == 0) {          // bias gradient: the 8 threads that share a row hold its partial sums
        float* part2 = PB ? p.part2 + (size_t)split * (p.Co + 1) : nullptr;
#pragma unroll
        for (int u = 0; u < MR; ++u) {
            float v = bs[u];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            const int o = o0 + row0 + 32 * u;
            if constexpr (PB) {
                float v2 = bs2[u];
                v2 += __shfl_xor(v2, 1); v2 += __shfl_xor(v2, 2); v2 += __shfl_xor(v2, 4);
                if ((tid & 7) == 0 && o < p.Co) { part[(size_t)o * (p.Ci + 1) + p.Ci] = v * p.pb_w2[o]; part2[o] = v2; }
            } else if ((tid & 7) == 0 && o < p.Co) part[(size_t)o * (p.Ci + 1) + p.Ci] = v;
        }
        if constexpr (PB) {
            float v = qs;
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            if (tid == 0 && o0 == 0) part2[p.Co] = v;
        }
    }
}

// second stage of the projection's own gradients (PB): one wave per entry, lanes stride over the splits, fixed order
__global__ __launch_bounds__(256) void channel_wgrad_pb_reduce_kernel(const float* __restrict__ part2, float* __restrict__ gw2, float* __restrict__ gb2,
                                                                      int Co, int nsplit) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e > Co) return;
    float v = 0.f;
    for (int k = lane; k < nsplit; k += 64) v += part2[(size_t)k * (Co + 1) + e];
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) {
        if (e < Co) gw2[e] = v;
        else if (gb2) gb2[0] = v;
    }
}


// Weight gradient with few input channels (CI <= 4): a thread owns four pixels at a time and accumulates its 16 output channels x
// (CI + 1) sums in registers over the pixels of its split (the + 1: the bias gradient); fixed-order reduction inside the workgroup
// (butterfly within a wave, the four waves through LDS in order), then the usual partials (split, Co, Ci + 1) for the reduce kernel.
template <int CI, bool BF>
__global__ __launch_bounds__(256) void channel_wgrad_few_in_kernel(ChannelWgradParams p, long long quads_per_split) {
    using T = typename IoElem<BF>::type;
    __shared__ float sred[4][16 * (CI + 1)];
    const int split = blockIdx.x, o0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long qrow = ((long long)p.P + 3) >> 2;                   // pixel quads per (batch entry, channel) row
    const long long qa = (long long)split * quads_per_split, qb = min(qa + quads_per_split, (long long)p.B * qrow);
    float acc[16][CI + 1];
#pragma unroll
    for (int o = 0; o < 16; ++o)
#pragma unroll
        for (int i = 0; i <= CI; ++i) acc[o][i] = 0.f;
    for (long long q = qa + tid; q < qb; q += 256) {
        const int b = (int)(q / qrow), px = (int)(4 * (q - (long long)b * qrow));
        const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * CI * p.P + px;
        const T* gb = reinterpret_cast<const T*>(p.gy) + ((size_t)b * p.Co + o0) * p.P + px;
        const bool full = px + 3 < p.P;
        float xv[CI][4];
#pragma unroll
        for (int i = 0; i < CI; ++i) {
            if (full) { const float4 v = io_ld4(xb + (size_t)i * p.P); xv[i][0] = v.x; xv[i][1] = v.y; xv[i][2] = v.z; xv[i][3] = v.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[i][e] = px + e < p.P ? io_widen(xb[(size_t)i * p.P + e]) : 0.f;
            }
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            float gv[4] = {0.f, 0.f, 0.f, 0.f};
            if (o0 + o < p.Co) {
                if (full) { const float4 v = io_ld4(gb + (size_t)o * p.P); gv[0] = v.x; gv[1] = v.y; gv[2] = v.z; gv[3] = v.w; }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) gv[e] = px + e < p.P ? io_widen(gb[(size_t)o * p.P + e]) : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < CI; ++i) acc[o][i] += (gv[0] * xv[i][0] + gv[1] * xv[i][1]) + (gv[2] * xv[i][2] + gv[3] * xv[i][3]);
            acc[o][CI] += (gv[0] + gv[1]) + (gv[2] + gv[3]);
        }
    }
#pragma unroll
    for (int o = 0; o < 16; ++o)
#pragma unroll
        for (int i = 0; i <= CI; ++i) {
            float v = acc[o][i];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            if (lane == 0) sred[wave][o * (CI + 1) + i] = v;
        }
    __syncthreads();
    if (tid < 16 * (CI + 1)) {
        const int o = tid / (CI + 1), i = tid % (CI + 1);
        const float v = ((sred[0][tid] + sred[1][tid]) + sred[2][tid]) + sred[3][tid];
        if (o0 + o < p.Co) p.part[((size_t)split * p.Co + o0 + o) * (CI + 1) + i] = v;
    }
}

// fixed-order sum of the split-K partials (deterministic): gw (Co, Ci), gb (Co).  32 consecutive elements x 8
// interleaved groups of splits per workgroup, the 8 group sums combined in order through LDS.
__global__ __launch_bounds__(256) void channel_wgrad_reduce_kernel(const float* part, float* gw, float* gb, int Co, int Ci, int nsplit,
                                                                   int accumulate) {
    __shared__ float sh[8][33];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const int n = Co * (Ci + 1);
    float acc = 0.f;
    if (e < n) {
        // 8 independent loads per round (the plain loop issued one load per iteration: a chain of nsplit / 8 memory latencies);
        // the order of the additions stays fixed
        int s = grp;
        for (; s + 56 < nsplit; s += 64) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = part[(size_t)(s + 8 * i) * n + e];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
        }
        for (; s < nsplit; s += 8) acc += part[(size_t)s * n + e];
    }
    sh[grp][el] = acc;
    __syncthreads();
    if (grp == 0 && e < n) {
        float t = sh[0][el];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += sh[g][el];
        const int o = e / (Ci + 1), i = e % (Ci + 1);
        // accumulate: the results are added to what gw / gb hold (a parameter's gradient buffer written in place)
        if (i < Ci) gw[(size_t)o * Ci + i] = accumulate == 1 ? gw[(size_t)o * Ci + i] + t : t;
        else if (gb) gb[o] = accumulate == 1 ? gb[o] + t : t;
    }
}

// split-K plan: ~1024 workgroups, each at least 4 chunks long; splits are whole chunks of one batch entry
static void wgrad_plan(int B, int Ci, int Co, long long P, int* nsplit, int* npc, int* cps, int* pk) {
    const int tiles = ((Co + CW_T - 1) / CW_T) * ((Ci + CW_T - 1) / CW_T);
    *pk = P >= 64 ? CWV_PK : CW_PK;
    *npc = (int)((P + *pk - 1) / *pk);
    const long long nchunks = (long long)B * *npc;
    long long want = (1024 + tiles - 1) / tiles;
    if (wgrad_split_shape(B, Ci, Co, P)) {             // K9-S: 128 x 128 weight tiles, two resident workgroups per CU = 512
        const int to = wgrad_split_rows(Co);
        const int tiles_s = ((Co + to - 1) / to) * ((Ci + CWS_T - 1) / CWS_T);
        want = (512 + tiles_s - 1) / tiles_s;
    }
    long long per = (nchunks + want - 1) / want;
    if (per < 4) per = 4;
    *cps = (int)per;
    *nsplit = (int)((nchunks + per - 1) / per);
}

long long channel_wgrad_ws_floats(int B, int Ci, int Co, long long P, int* nsplit_out) {
    int nsplit, npc, cps, pk;
    wgrad_plan(B, Ci, Co, P, &nsplit, &npc, &cps, &pk);
    if (nsplit_out) *nsplit_out = nsplit;
    return (long long)nsplit * Co * (long long)(Ci + 1);
}

int launch_channel_wgrad(const void* gy, const void* x, float* gw, float* gb, float* ws, int B, int Ci, int Co, long long P,
                         int act_x, int bf16, hipStream_t s) {
    return launch_channel_wgrad2(gy, x, nullptr, Ci, gw, gb, ws, B, Ci, Co, P, act_x, 0, bf16, s);
}

int launch_channel_wgrad_vh(const void* gy, const float* vh_x, const float* vh_w, const float* vh_b, int vh_ci, float* gw, float* gb, float* ws,
                            int B, int Ci, int Co, long long P, int act_x, hipStream_t s, int accumulate) {
    if (Ci > 32 || Ci < 5 || vh_ci < 1 || vh_ci > 3 || P < 64 || !act_x || (long long)Co * P >= (1LL << 29) || (long long)B * ((P + 31) / 32) > 0x7fffffffLL) {
        set_error("channel_wgrad: the virtual-input form takes 5 .. 32 virtual channels of <= 3 real ones, read through the GELU, >= 64 pixels");
        return -2;
    }
    ChannelWgradParams p;
    p.gy = gy; p.x = vh_x; p.x2 = nullptr; p.C1 = Ci; p.part = ws; p.B = B; p.Ci = Ci; p.Co = Co; p.P = (int)P; p.act_x = 1;
    p.pm = pix_map(PixelWindow(), P);
    p.vh_x = vh_x; p.vh_w = vh_w; p.vh_b = vh_b; p.vh_ci = vh_ci;
    p.pb_w2 = nullptr; p.pb_g = nullptr; p.part2 = nullptr;
    int npc, cps, pk;
    wgrad_plan(B, Ci, Co, P, &p.nsplit, &npc, &cps, &pk);
    p.span = (long long)cps * pk;
    p.rev = next_sweep_reversed(SWEEP_K9);
    const int tiles = ((Co + CW_T - 1) / CW_T) * ((Ci + CW_T - 1) / CW_T);
    {
        ProfScope prof("uno::channel_wgrad_vec_kernel", 4.0 * B * (double)P * (vh_ci + Co), s);
        hipLaunchKernelGGL((channel_wgrad_vec_kernel<true, false, 2, true>), dim3(8 * tiles * ((p.nsplit + 7) / 8)), dim3(256), 0, s, p, npc, cps);
    }
    hipLaunchKernelGGL(channel_wgrad_reduce_kernel, dim3((Co * (Ci + 1) + 31) / 32), dim3(256), 0, s, ws, gw, gb, Co, Ci, p.nsplit, accumulate ? 1 : 0);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("channel_wgrad launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

bool channel_wgrad_pb_applies(int B, int Ci, int Co, int C1, long long P) {
    int nsplit, npc, cps, pk;
    if (B < 1 || P < 64) return false;
    wgrad_plan(B, Ci, Co, P, &nsplit, &npc, &cps, &pk);
    return wgrad_split_shape(B, Ci, Co, P) && pk == CWV_PK && Ci > 4 && (C1 == Ci || C1 % 32 == 0);
}
long long channel_wgrad_pb_ws_floats(int B, int Ci, int Co, long long P) {
    int nsplit;
    const long long n = channel_wgrad_ws_floats(B, Ci, Co, P, &nsplit);
    return n + (long long)nsplit * (Co + 1);
}

int launch_channel_wgrad2(const void* gy, const void* x, const void* x2, int C1, float* gw, float* gb, float* ws, int B, int Ci, int Co,
                          long long P, int act_x, int accumulate, int bf16, hipStream_t s, const PixelWindow& win, const WgradProjectedBack& pb) {
    const bool windowed = win.cols != 0;
    if (const char* why = pix_window_error(win, P)) { set_error("channel_wgrad: %s", why); return -2; }
    const long long PSl = windowed ? win.plane : P;
    if ((long long)(Ci > Co ? Ci : Co) * PSl >= (1LL << 29) || (long long)B * ((P + 31) / 32) > 0x7fffffffLL) {
        set_error("channel_wgrad: tensor too large (channels * pixels must stay below 2^29)");
        return -2;
    }
    if (x2 && (P < 64 || C1 < CW_T || C1 >= Ci || C1 % CW_T)) {
        set_error("channel_wgrad: a two-source call needs >= 64 pixels and a split at a multiple of %d inside (0, Ci) (got %d of %d)", CW_T, C1, Ci);
        return -2;
    }
    ChannelWgradParams p;
    p.gy = gy; p.x = x; p.x2 = x2; p.C1 = x2 ? C1 : Ci; p.part = ws; p.B = B; p.Ci = Ci; p.Co = Co; p.P = (int)P; p.act_x = act_x ? 1 : 0;
    p.pm = pix_map(win, P);
    p.vh_x = nullptr; p.vh_w = nullptr; p.vh_b = nullptr; p.vh_ci = 0;
    int npc, cps, pk;
    wgrad_plan(B, Ci, Co, P, &p.nsplit, &npc, &cps, &pk);
    p.pb_w2 = pb.w2; p.pb_g = pb.g; p.part2 = ws + (size_t)p.nsplit * Co * (Ci + 1);
    if (pb.w2 && (!pb.g || !pb.gw2 || bf16 || accumulate == 3 || !channel_wgrad_pb_applies(B, Ci, Co, x2 ? C1 : Ci, P))) {
        set_error("channel_wgrad: the projected-back gradient goes with the float32 split kernel (both stages)");
        return -3;
    }
    if (windowed && (pk != CWV_PK || (Ci <= 4 && !act_x))) { set_error("channel_wgrad: the pixel window goes with the vector / split kernels (>= 64 pixels, > 4 input channels)"); return -2; }
    p.span = (long long)cps * pk;
    p.rev = next_sweep_reversed(SWEEP_K9);
    const int tiles = ((Co + CW_T - 1) / CW_T) * ((Ci + CW_T - 1) / CW_T);
    if (Ci <= 4 && !act_x && P >= 1024) {
        const long long quads = (long long)B * ((P + 3) / 4), qps = (quads + p.nsplit - 1) / p.nsplit;
        {
            ProfScope prof("uno::channel_wgrad_few_in_kernel", (bf16 ? 2.0 : 4.0) * B * (double)P * (Ci + Co), s);
            const dim3 grid((unsigned)p.nsplit, (unsigned)((Co + 15) / 16));
#define UNO_CWF(C) do { if (bf16) hipLaunchKernelGGL((channel_wgrad_few_in_kernel<C, true>), grid, dim3(256), 0, s, p, qps); \
                        else hipLaunchKernelGGL((channel_wgrad_few_in_kernel<C, false>), grid, dim3(256), 0, s, p, qps); } while (0)
            if (Ci == 1) UNO_CWF(1); else if (Ci == 2) UNO_CWF(2); else if (Ci == 3) UNO_CWF(3); else UNO_CWF(4);
#undef UNO_CWF
        }
        const int nf = Co * (Ci + 1);
        if (accumulate != 3)
            hipLaunchKernelGGL(channel_wgrad_reduce_kernel, dim3((nf + 31) / 32), dim3(256), 0, s, ws, gw, gb, Co, Ci, p.nsplit, accumulate);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("channel_wgrad launch: %s", hipGetErrorString(e)); return -5; }
        return 0;
    }
    const bool split_form = wgrad_split_shape(B, Ci, Co, P) && pk == CWV_PK && (!x2 || C1 % 32 == 0);
    {
        ProfScope prof(split_form ? "uno::channel_wgrad_split_kernel" : pk == CWV_PK ? "uno::channel_wgrad_vec_kernel" : "uno::channel_wgrad_kernel",
                       (bf16 ? 2.0 : 4.0) * B * (double)P * (Ci + Co + (pb.w2 ? 1 : 0)), s);
        const dim3 gv(8 * tiles * ((p.nsplit + 7) / 8));
        if (split_form) {
            const int to = wgrad_split_rows(Co);
            const int tiles_s = ((Co + to - 1) / to) * ((Ci + CWS_T - 1) / CWS_T);
            const dim3 gs(8 * tiles_s * ((p.nsplit + 7) / 8));
#define UNO_CWS(A_, M_) do { if (bf16) hipLaunchKernelGGL((channel_wgrad_split_kernel<A_, M_, true>), gs, dim3(256), 0, s, p, npc, cps); \
                             else hipLaunchKernelGGL((channel_wgrad_split_kernel<A_, M_, false>), gs, dim3(256), 0, s, p, npc, cps); } while (0)
#define UNO_CWS_PB(A_, M_) hipLaunchKernelGGL((channel_wgrad_split_kernel<A_, M_, false, true>), gs, dim3(256), 0, s, p, npc, cps)
            if (pb.w2) {
                if (to == CWS_T) { if (act_x) UNO_CWS_PB(true, 4); else UNO_CWS_PB(false, 4); }
                else { if (act_x) UNO_CWS_PB(true, 2); else UNO_CWS_PB(false, 2); }
            }
            else if (to == CWS_T) { if (act_x) UNO_CWS(true, 4); else UNO_CWS(false, 4); }
            else { if (act_x) UNO_CWS(true, 2); else UNO_CWS(false, 2); }
#undef UNO_CWS_PB
#undef UNO_CWS
        } else if (pk == CWV_PK && Ci <= 32) {           // (one tile of input channels: the narrow form)
            if (act_x) { if (bf16) hipLaunchKernelGGL((channel_wgrad_vec_kernel<true, true, 2>), gv, dim3(256), 0, s, p, npc, cps);
                         else hipLaunchKernelGGL((channel_wgrad_vec_kernel<true, false, 2>), gv, dim3(256), 0, s, p, npc, cps); }
            else { if (bf16) hipLaunchKernelGGL((channel_wgrad_vec_kernel<false, true, 2>), gv, dim3(256), 0, s, p, npc, cps);
                   else hipLaunchKernelGGL((channel_wgrad_vec_kernel<false, false, 2>), gv, dim3(256), 0, s, p, npc, cps); }
        } else if (pk == CWV_PK && act_x) {
            if (bf16) hipLaunchKernelGGL((channel_wgrad_vec_kernel<true, true>), gv, dim3(256), 0, s, p, npc, cps);
            else hipLaunchKernelGGL((channel_wgrad_vec_kernel<true, false>), gv, dim3(256), 0, s, p, npc, cps);
        } else if (pk == CWV_PK) {
            if (bf16) hipLaunchKernelGGL((channel_wgrad_vec_kernel<false, true>), gv, dim3(256), 0, s, p, npc, cps);
            else hipLaunchKernelGGL((channel_wgrad_vec_kernel<false, false>), gv, dim3(256), 0, s, p, npc, cps);
        } else if (bf16) {
            hipLaunchKernelGGL(channel_wgrad_kernel<true>, dim3(tiles, p.nsplit), dim3(256), 0, s, p, npc, cps);
        } else {
            hipLaunchKernelGGL(channel_wgrad_kernel<false>, dim3(tiles, p.nsplit), dim3(256), 0, s, p, npc, cps);
        }
    }
    const int n = Co * (Ci + 1);
    if (accumulate != 3)        // 3: the partial sums stay in ws; launch_channel_wgrad_finish sums any number of such blocks later
        hipLaunchKernelGGL(channel_wgrad_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, s, ws, gw, gb, Co, Ci, p.nsplit, accumulate);
    if (pb.w2) hipLaunchKernelGGL(channel_wgrad_pb_reduce_kernel, dim3((Co + 1 + 3) / 4), dim3(256), 0, s, p.part2, pb.gw2, pb.gb2, Co, p.nsplit);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("channel_wgrad launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

// second stage alone: nparts consecutive (Co, Ci + 1) blocks of partial sums (the ws of one or more stage-1 calls with
// accumulate = 3, laid out one after the other) -> gw, gb, in the fixed order of the blocks
int launch_channel_wgrad_finish(const float* parts, float* gw, float* gb, int Ci, int Co, long long nparts, int accumulate, hipStream_t s) {
    if (nparts < 1 || nparts > 0x7fffffffLL) { set_error("channel_wgrad_finish: %lld partial blocks", nparts); return -2; }
    const int n = Co * (Ci + 1);
    hipLaunchKernelGGL(channel_wgrad_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, s, parts, gw, gb, Co, Ci, (int)nparts, accumulate ? 1 : 0);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("channel_wgrad_finish launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
