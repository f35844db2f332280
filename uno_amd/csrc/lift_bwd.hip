// K15 - the backward pass of the lift (reference darcy_flow_uno2d.py:98-107: fc_n1, GELU, fc0, GELU, pad) in ONE kernel per pixel tile.
//
//   h = w1 x + b1 (Cm = 32 channels from Cin <= 3),  a = gelu(h),  z = w0 a + b0 (Co = 64),  act = pad(gelu(z))        forward
//   gz = gelu'(z) * g[..., :H, :W]                      (g: the gradient of the padded activation)
//   gh = gelu'(h) * (w0^T gz),  gw1 = sum_px gh x^T,  gb1 = sum_px gh      (x is data: gh has no other consumer and is not stored)
//   gw0 = sum_px gz a^T,  gb0 = sum_px gz               -> one (Co, Cm + 1) block of partial sums per workgroup, summed by K9's second stage
//
// The four-kernel form (capi.hip, uno_lift_backward) writes gz (64 channels) once and reads it twice: 2.2 GB of the pass's 3.8 GB.
// Here gz lives in LDS only: a workgroup of 256 threads owns 128 pixels, recomputes a and z, and runs the three small GEMMs
// (z: 64 x 32 x 128, gh: 32 x 64 x 128, gw0: 64 x 32 over its 128 pixels) on v_mfma_f32_16x16x4_f32 from LDS-resident operands.
// HBM traffic: x (12 B per pixel), g (256 B) and 9 KB of partial sums per TPW tiles.
//
// LDS (pitch TS = 148 floats: rows 16-byte aligned; 148 mod 64 = 20 makes the k-contiguous fragment reads of the weight-gradient
// GEMM - lane (row r16, k kk) at r16 * 148 + kk - hit 64 distinct banks):
//   sA [32][TS]  a = gelu(h)   (operand of the z and gw0 GEMMs; afterwards the staging area of gh's row-wise stores)
//   sZ [64][TS]  g, then gz in place (the lane that reads g[o][px] writes gz[o][px])
//   sW [32][80]  w0 as [m][o];  sWT [64][48]  w0 as [o][m];  sVH [32] (w1 row, b1)
// 78 KB: two workgroups per CU.  MFMA operand convention as everywhere in this library: D[i][j] = sum_k A[i][k] B[k][j], a lane
// (r16, kk) supplies A[i = r16][k = kk] and B[k = kk][j = r16] and receives D[i = 4 kk + r][j = r16], r = 0..3.
#include "uno_common.h"
#include <cstdlib>

namespace uno {

constexpr int LB_PT = 128;          // pixels per tile
constexpr int LB_TS = 148;          // LDS row pitch (floats)
constexpr int LB_CM = 32, LB_CO = 64;
constexpr int LB_WS = 80, LB_WTS = 48;       // k rows 16 banks apart for the four k-lanes of a fragment read
#ifndef UNO_LB_TPW
#define UNO_LB_TPW 8
#endif
constexpr int LB_TPW = UNO_LB_TPW;  // pixel tiles per workgroup (one block of partial sums per workgroup)

// Pixel SLOTS.  Both kernels walk the H x W domain as H rows of W4 = W rounded up to a multiple of 4 slots: a thread's four slots are
// four columns of ONE row in every tensor - dense (pitch W: x, gh) and padded (pitch Wp: g, act) alike - so no quad straddles a row
// end, padded rows are read and written from 8 / 16-byte aligned addresses, and the one to three slots past a row's end are masked.
// (Flat pixel quads, the first version, straddled row ends in the padded tensors: every quad of a tile with a row end was loaded twice.)
struct LiftGeom {
    int H, W, W4, nslot, npt;           // nslot = H * W4, npt = tiles of 128 slots
    int skip_d, skip_p;                 // offset of slot s in row r: s + r * skip (dense: W - W4 <= 0, padded: Wp - W4)
    int Pd, Pp;                         // plane sizes: H * W, Hp * Wp
    unsigned magic;                     // ceil(2^40 / W4)
};
static LiftGeom lift_geom(int H, int W, int Hp, int Wp, int slots_per_row = 0) {
    LiftGeom g;
    g.H = H; g.W = W; g.W4 = slots_per_row ? slots_per_row : ((W + 3) & ~3); g.nslot = H * g.W4; g.npt = (g.nslot + 127) / 128;
    g.skip_d = W - g.W4; g.skip_p = Wp - g.W4; g.Pd = H * W; g.Pp = Hp * Wp;
    g.magic = (unsigned)(((1ULL << 40) + g.W4 - 1) / (unsigned long long)g.W4);
    return g;
}
__device__ __forceinline__ int lift_clamped(int off, int plane_len) { return max(min(off, plane_len - 4), 0); }
// this thread's quad of a tile: row, valid slots (0 .. 4), offsets in a dense and in a padded plane
struct LiftQuad { int nv, od, op, row_ok, last, shd, shp; };         // row_ok: the row exists; last: the quad is its row's last one;
                                                                     // od / op: CLAMPED offsets, shd / shp: elements the loaded piece must be shifted by
__device__ __forceinline__ LiftQuad lift_quad(const LiftGeom& g, int s0, int s) {
    const int r0 = (int)(((unsigned long long)(unsigned)s0 * g.magic) >> 40);      // (uniform) row of the tile's first slot
    const int r = r0 + (s >= (r0 + 1) * g.W4 ? 1 : 0);                             // a tile of 128 slots touches two rows at most (W4 >= 260)
    const int col = s - r * g.W4;
    const int od = s + r * g.skip_d, op = s + r * g.skip_p, odc = lift_clamped(od, g.Pd), opc = lift_clamped(op, g.Pp);
    return LiftQuad{r < g.H ? min(max(g.W - col, 0), 4) : 0, odc, opc, r < g.H ? 1 : 0, col == g.W4 - 4 ? 1 : 0, od - odc, op - opc};
}
// four floats at plane[off ..] from an address clamped into the plane (the last quads of a plane's last row), shifted back into place
// The RAW piece is returned: the shift is applied where the value is consumed (lift_fix) - applied here it used the loaded registers at
// once and the wave waited for every load right after issuing it (the first phase of both kernels: 6 k - 7 k cycles per tile).
__device__ __forceinline__ float4 lift_fix(const float4& v, int sh) {
    float t0 = v.x, t1 = v.y, t2 = v.z, t3 = v.w;
    if (sh & 1) { t0 = t1; t1 = t2; t2 = t3; }
    if (sh & 2) { t0 = t2; t1 = t3; }
    return make_float4(t0, t1, t2, t3);           // (elements past the plane's end hold stale values: the caller masks by nv)
}
__device__ __forceinline__ float4 lift_mask(const float4& v, int nv) {
    return make_float4(nv > 0 ? v.x : 0.f, nv > 1 ? v.y : 0.f, nv > 2 ? v.z : 0.f, nv > 3 ? v.w : 0.f);
}

struct LiftBwdParams {
    const float* x;         // (B, Cin, P)
    const float* w1;        // (32, Cin)
    const float* b1;        // (32) or nullptr
    const float* w0;        // (64, 32)
    const float* b0;        // (64) or nullptr
    const float* g;         // (B, 64, Hp, Wp)
    const float* g2;        // optional second gradient of the padded activation, same planes, valid on the domain (the skip connection's
                            // other consumer: reference darcy_flow_uno2d.py:122-127 feeds x_fc0 to fc1 as well); nullptr = none
    float* part;            // (B * wg_per_batch, 64, 33): fc0's weight / bias gradient, one block of partial sums per workgroup
    float* part1;           // (B * wg_per_batch, 32, Cin + 1): fc_n1's - gh = gelu'(h) (w0^T gz) never leaves the kernel either
    int B, Cin;
    LiftGeom geo;
    unsigned long long* stamps;     // development (-DUNO_LB_DEV): per-phase cycles of every wave, 8 values each
    int rev;                        // alternating sweep direction (uno_common.h): tile groups and batch entries in descending order
};

__device__ __forceinline__ float4 lb_vh(const float4& t, const float4* q) {       // t = (w[c][0..2], b[c]); q: the real channels at 4 pixels
    return make_float4(fmaf(t.z, q[2].x, fmaf(t.y, q[1].x, fmaf(t.x, q[0].x, t.w))), fmaf(t.z, q[2].y, fmaf(t.y, q[1].y, fmaf(t.x, q[0].y, t.w))),
                       fmaf(t.z, q[2].z, fmaf(t.y, q[1].z, fmaf(t.x, q[0].z, t.w))), fmaf(t.z, q[2].w, fmaf(t.y, q[1].w, fmaf(t.x, q[0].w, t.w))));
}

// gelu(x) and gelu'(x) from one erf
__device__ __forceinline__ void lb_gelu_both(float x, float& g, float& d) {
    const float cdf = 0.5f * (1.f + uno_erf(x * 0.70710678118654752440f));
    g = x * cdf;
    d = fmaf(x, 0.39894228040143267794f * __expf(-0.5f * x * x), cdf);
}

__global__ __launch_bounds__(256, 2) void lift_backward_kernel(LiftBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lb_smem[];
    float* sA = lb_smem;                            // [32][TS]
    float* sZ = sA + LB_CM * LB_TS;                 // [64][TS]
    float* sW = sZ + LB_CO * LB_TS;                 // [32][68]   w0[o][m] at [m][o]
    float* sWT = sW + LB_CM * LB_WS;                // [64][36]   w0[o][m] at [o][m]
    float4* sVH = reinterpret_cast<float4*>(sWT + LB_CO * LB_WTS);      // [32]
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = sweep_y(p.rev), bxs = sweep_x(p.rev);
    const LiftGeom& G = p.geo;

    // weights: once per workgroup
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = tid + 256 * u, o = e >> 5, m = e & 31;
        const float v = p.w0[e];
        sW[m * LB_WS + o] = v;
        sWT[o * LB_WTS + m] = v;
    }
    if (tid < LB_CM) {
        const float* wr = p.w1 + tid * p.Cin;
        sVH[tid] = make_float4(wr[0], p.Cin > 1 ? wr[1] : 0.f, p.Cin > 2 ? wr[2] : 0.f, p.b1 ? p.b1[tid] : 0.f);
    }
    const float b0v = p.b0 ? p.b0[16 * wave + r16] : 0.f;          // z's channel of this lane in the MFMA layout
    const float* xb = p.x + (size_t)b * p.Cin * G.Pd;
    const float* gb = p.g + (size_t)b * LB_CO * G.Pp;
    const float* gb2 = p.g2 ? p.g2 + (size_t)b * LB_CO * G.Pp : nullptr;

    f32x4 acc3[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};       // gw0 partial: D[o = 16 wave + 4 kk + r][m = 16 t + r16], over all tiles of this workgroup
    float bsum = 0.f;                                               // gb0 partial of channel 16 wave + r16 (this lane's pixels)

#ifdef UNO_LB_DEV
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
#define LB_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define LB_STAMP(i) do { } while (0)
#endif
    const int t_begin = bxs * LB_TPW, t_end = min(t_begin + LB_TPW, G.npt);
    // a tile's global loads: this thread's quad (slots 4 (tid & 31) ..) of the real channels and of rows (tid >> 5) + 8 u of g - straight-
    // line code (the guarded form of the first version, branches around scalar tails with a wait inside each, serialised the eight row
    // loads: 17 k of a tile's 38 k cycles).  The NEXT tile's loads are issued as soon as the current tile's values are in LDS and arrive
    // while its GEMMs run (two workgroups per CU cannot hide them otherwise).
    float4 xq[3], gq[8];
    float4 xc[3];               // the CURRENT tile's real channels (masked): operand of fc_n1's weight gradient at the end of the tile
    float g1[4][4];             // fc_n1: partial sums of rows (tid >> 5) + 8 u against the three real channels, and of the rows themselves
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) g1[u][k] = 0.f;
    LiftQuad cur = {0, 0, 0, 0, 0, 0, 0}, nxt = {0, 0, 0, 0, 0, 0, 0};
    auto load_tile = [&](int tile) {
        const int s0 = tile * LB_PT;
        nxt = lift_quad(G, s0, s0 + (tid & 31) * 4);
#pragma unroll
        for (int k = 0; k < 3; ++k) xq[k] = io_ld4(xb + (size_t)min(k, p.Cin - 1) * G.Pd + nxt.od);
#pragma unroll
        for (int u = 0; u < 8; ++u) gq[u] = io_ld4(gb + (size_t)((tid >> 5) + 8 * u) * G.Pp + nxt.op);
    };
    if (t_begin < t_end) load_tile(t_begin);
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        cur = nxt;
#pragma unroll
        for (int k = 0; k < 3; ++k) { xq[k] = lift_fix(xq[k], cur.shd); xc[k] = lift_mask(xq[k], cur.nv); }
        // ---- phase 0: a = gelu(h) -> sA (rows (tid >> 5) + 8 u at quad tid & 31), g -> sZ (row-wise, 16-byte pieces); gelu'(h) of the
        // same elements stays in registers for the end of the tile, where this thread stores exactly these elements of gh
        float4 dh[4];
        {
            const int q4 = (tid & 31) * 4;
            const float vm0 = cur.nv > 0 ? 1.f : 0.f, vm1 = cur.nv > 1 ? 1.f : 0.f, vm2 = cur.nv > 2 ? 1.f : 0.f, vm3 = cur.nv > 3 ? 1.f : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = (tid >> 5) + 8 * u;
                const float4 h = lb_vh(sVH[m], xq);
                float4 a4;
                lb_gelu_both(h.x, a4.x, dh[u].x); lb_gelu_both(h.y, a4.y, dh[u].y); lb_gelu_both(h.z, a4.z, dh[u].z); lb_gelu_both(h.w, a4.w, dh[u].w);
                // slots past a row's end contribute nothing to the weight gradient: a = 0 there
                *reinterpret_cast<float4*>(sA + m * LB_TS + q4) = make_float4(a4.x * vm0, a4.y * vm1, a4.z * vm2, a4.w * vm3);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(sZ + ((tid >> 5) + 8 * u) * LB_TS + q4) = lift_mask(lift_fix(gq[u], cur.shp), cur.nv);
        }
        if (gb2) {
            // second gradient tensor: its eight row pieces of THIS tile go through the registers the first one just left, and are added
            // to sZ in place (same thread, same elements) once the first half of the z GEMM has covered their latency; the next tile's
            // loads follow.  (A second prefetched register set does not fit: 252 of 256 registers at two workgroups per CU.)
#pragma unroll
            for (int u = 0; u < 8; ++u) gq[u] = io_ld4(gb2 + (size_t)((tid >> 5) + 8 * u) * G.Pp + cur.op);
        } else if (tile + 1 < t_end) load_tile(tile + 1);
        LB_STAMP(0);
        __syncthreads();
        LB_STAMP(1);
        // ---- z = w0 a: D[px = 16 mt + 4 kk + r][o = 16 wave + r16], in two halves of four pixel tiles: the gelu' epilogue of the first
        // half (16 evaluations per lane, ~370 VALU instructions) is placed BETWEEN the 32 MFMAs of the second half - inside one wave's
        // program order independent VALU work runs while the matrix pipe is busy
        f32x4 acc1[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc1[mt] = f32x4{0, 0, 0, 0};
        float wv[LB_CM / 4];
#pragma unroll
        for (int ks = 0; ks < LB_CM / 4; ++ks) wv[ks] = sW[(4 * ks + kk) * LB_WS + 16 * wave + r16];
#pragma unroll
        for (int ks = 0; ks < LB_CM / 4; ++ks) {
            const float* arow = sA + (4 * ks + kk) * LB_TS + r16;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc1[mt] = mfma16(arow[16 * mt], wv[ks], acc1[mt]);
        }
        LB_STAMP(2);
        if (gb2) {
            const int q4 = (tid & 31) * 4;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float4* dst = reinterpret_cast<float4*>(sZ + ((tid >> 5) + 8 * u) * LB_TS + q4);
                const float4 a = *dst, b2 = lift_mask(lift_fix(gq[u], cur.shp), cur.nv);
                *dst = make_float4(a.x + b2.x, a.y + b2.y, a.z + b2.z, a.w + b2.w);
            }
            if (tile + 1 < t_end) load_tile(tile + 1);
            __syncthreads();                // the epilogue below reads other threads' rows of sZ
        }
        float* const zrow = sZ + (16 * wave + r16) * LB_TS + 4 * kk;       // gz = gelu'(z + b0) * g, in place in sZ (this wave's rows only)
        auto gz_tile = [&](int mt) {
            const float4 g4 = *reinterpret_cast<const float4*>(zrow + 16 * mt);
            const float4 gz = make_float4(uno_dgelu(acc1[mt][0] + b0v) * g4.x, uno_dgelu(acc1[mt][1] + b0v) * g4.y,
                                          uno_dgelu(acc1[mt][2] + b0v) * g4.z, uno_dgelu(acc1[mt][3] + b0v) * g4.w);
            *reinterpret_cast<float4*>(zrow + 16 * mt) = gz;
            bsum += (gz.x + gz.y) + (gz.z + gz.w);
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < LB_CM / 4; ++ks) {
            const float* arow = sA + (4 * ks + kk) * LB_TS + r16;
#pragma unroll
            for (int mt = 4; mt < 8; ++mt) acc1[mt] = mfma16(arow[16 * mt], wv[ks], acc1[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gz_tile(mt);
        __builtin_amdgcn_sched_group_barrier(0x100, 36, 0);             // the second half's operand reads, the first half's g
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);          // ~1/32 of the epilogue's VALU
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 4; mt < 8; ++mt) gz_tile(mt);
        LB_STAMP(3);
        __syncthreads();
        LB_STAMP(1);
        // ---- gh = w0^T gz: wave (wp, wm) = (pixel half, channel group): D[px = 64 wp + 16 mt + 4 kk + r][m = 16 wm + r16]
        const int wp = wave >> 1, wm = wave & 1;
        f32x4 acc2[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc2[mt] = f32x4{0, 0, 0, 0};
#pragma unroll 4
        for (int ks = 0; ks < LB_CO / 4; ++ks) {
            const float wv = sWT[(4 * ks + kk) * LB_WTS + 16 * wm + r16];
            const float* zrow = sZ + (4 * ks + kk) * LB_TS + 64 * wp + r16;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc2[mt] = mfma16(zrow[16 * mt], wv, acc2[mt]);
        }
        LB_STAMP(4);
        // ---- gw0 += gz a^T over this tile's pixels: A[i = o][k = px] = sZ row 16 wave + r16, B[k = px][j = m] = sA row 16 t + r16
        {
            const float* zr = sZ + (16 * wave + r16) * LB_TS + kk;
            const float* a0 = sA + r16 * LB_TS + kk;
            const float* a1 = sA + (16 + r16) * LB_TS + kk;
#pragma unroll 8
            for (int ks = 0; ks < LB_PT / 4; ++ks) {
                const float zv = zr[4 * ks];
                acc3[0] = mfma16(zv, a0[4 * ks], acc3[0]);
                acc3[1] = mfma16(zv, a1[4 * ks], acc3[1]);
            }
        }
        LB_STAMP(5);
        __syncthreads();                // every wave is done with a: sA becomes the staging area of gh
        LB_STAMP(1);
        // w0^T gz -> sA in the row layout
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            *reinterpret_cast<float4*>(sA + (16 * wm + r16) * LB_TS + 64 * wp + 16 * mt + 4 * kk) = make_float4(acc2[mt][0], acc2[mt][1], acc2[mt][2], acc2[mt][3]);
        LB_STAMP(6);
        __syncthreads();
        LB_STAMP(1);
        // gh = gelu'(h) * (w0^T gz) in the row layout, thread -> (row (tid >> 5) + 8 u, quad tid & 31): not stored - x is data, gh's only
        // consumer is fc_n1's weight gradient gw1 = sum gh x^T, gb1 = sum gh, accumulated here (3 real channels: 4 FMAs per value)
        {
            const int q4 = (tid & 31) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = (tid >> 5) + 8 * u;
                float4 v = *reinterpret_cast<const float4*>(sA + m * LB_TS + q4);
                v = lift_mask(make_float4(v.x * dh[u].x, v.y * dh[u].y, v.z * dh[u].z, v.w * dh[u].w), cur.nv);
#pragma unroll
                for (int k = 0; k < 3; ++k) g1[u][k] = fmaf(v.x, xc[k].x, fmaf(v.y, xc[k].y, fmaf(v.z, xc[k].z, fmaf(v.w, xc[k].w, g1[u][k]))));
                g1[u][3] += (v.x + v.y) + (v.z + v.w);
            }
        }
        LB_STAMP(7);
        __syncthreads();                // before the next tile overwrites sA / sZ
        LB_STAMP(1);
    }
#ifdef UNO_LB_DEV
    if (p.stamps && lane == 0) {
        unsigned long long* o_ = p.stamps + (((size_t)b * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) o_[i] = tph[i];
    }
#endif
    // partial sums of this workgroup: (64, 33) block, bias in column 32
    float* part = p.part + ((size_t)b * gridDim.x + bxs) * (LB_CO * (LB_CM + 1));
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(16 * wave + 4 * kk + r) * (LB_CM + 1) + 16 * t + r16] = acc3[t][r];
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (kk == 0) part[(16 * wave + r16) * (LB_CM + 1) + LB_CM] = bsum;
    // fc_n1: the 32 lanes of a half wave hold the same rows; (32, Cin + 1) block, bias in column Cin
    float* part1 = p.part1 + ((size_t)b * gridDim.x + bxs) * (LB_CM * (p.Cin + 1));
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = g1[u][k];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
            if ((tid & 31) == 0) {
                const int m = (tid >> 5) + 8 * u;
                if (k < p.Cin) part1[m * (p.Cin + 1) + k] = v;
                else if (k == 3) part1[m * (p.Cin + 1) + p.Cin] = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------ K16: the forward pass, same scheme
// act = zero-pad(gelu(w0 gelu(w1 x + b1) + b0)): per 128-slot tile a = gelu(h) -> LDS, z = w0 a on the MFMA in two halves with the first
// half's GELU between the second half's MFMAs, the 64 x 128 result through LDS into row-wise 16-byte stores.  The slots of a row cover
// the WHOLE padded row (Wp rounded down to a multiple of 4; the lane of a row's last quad adds the one to three zeros that remain): the
// kernel writes the padding columns itself - whole 128-byte lines instead of a row's ragged end followed, in another launch, by the
// 100-byte strip next to it (first version of this kernel, flat pixels + clear_border: 345 us at 2.2 TB/s, like the generic K8 tile).
// Only the rows below the domain are left to clear_border (contiguous).
struct LiftFwdParams {
    const float* x; const float* w1; const float* b1; const float* w0; const float* b0;
    float* act;             // (B, 64, Hp, Wp)
    int B, Cin, tail;       // tail: columns of a padded row behind its last quad (Wp mod 4)
    LiftGeom geo;
    unsigned long long* stamps;     // development (-DUNO_LB_DEV)
    int rev;                // alternating sweep direction (uno_common.h)
};

__global__ __launch_bounds__(256, 2) void lift_forward_kernel(LiftFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lb_smem[];
    float* sA = lb_smem;                            // [32][TS]  a = gelu(h)
    float* sO = sA + LB_CM * LB_TS;                 // [64][TS]  gelu(z), row layout
    float* sW = sO + LB_CO * LB_TS;                 // [32][80]  w0 as [m][o]
    float4* sVH = reinterpret_cast<float4*>(sW + LB_CM * LB_WS);
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = sweep_y(p.rev), bxs = sweep_x(p.rev);
    const LiftGeom& G = p.geo;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = tid + 256 * u;
        sW[(e & 31) * LB_WS + (e >> 5)] = p.w0[e];
    }
    if (tid < LB_CM) {
        const float* wr = p.w1 + tid * p.Cin;
        sVH[tid] = make_float4(wr[0], p.Cin > 1 ? wr[1] : 0.f, p.Cin > 2 ? wr[2] : 0.f, p.b1 ? p.b1[tid] : 0.f);
    }
    const float b0v = p.b0 ? p.b0[16 * wave + r16] : 0.f;
    const float* xb = p.x + (size_t)b * p.Cin * G.Pd;
    float* ab = p.act + (size_t)b * LB_CO * G.Pp;
    // the real channels of a tile are requested TWO tiles ahead (register sets A / B in turn): a tile of this kernel lasts ~9 k cycles,
    // less than a load that misses L2 takes under load - with one tile of distance the first phase waited for them (5.8 k of 15.3 k cycles)
    float4 xqA[3], xqB[3];
    LiftQuad qdA = {0, 0, 0, 0, 0, 0, 0}, qdB = {0, 0, 0, 0, 0, 0, 0};
    auto load_x = [&](int tile, float4* xq, LiftQuad& qd) {
        const int s0 = tile * LB_PT;
        qd = lift_quad(G, s0, s0 + (tid & 31) * 4);
#pragma unroll
        for (int k = 0; k < 3; ++k) xq[k] = io_ld4(xb + (size_t)min(k, p.Cin - 1) * G.Pd + qd.od);
    };
#ifdef UNO_LB_DEV
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
#endif
    const int t_begin = bxs * LB_TPW, t_end = min(t_begin + LB_TPW, G.npt);
    if (t_begin < t_end) load_x(t_begin, xqA, qdA);
    if (t_begin + 1 < t_end) load_x(t_begin + 1, xqB, qdB);
    __syncthreads();
    auto do_tile = [&](int tile, float4* xq, LiftQuad& qd) {
        const LiftQuad cur = qd;
        {
            const int q4 = (tid & 31) * 4;
            const float4 q[3] = {lift_mask(lift_fix(xq[0], cur.shd), cur.nv), lift_mask(lift_fix(xq[1], cur.shd), cur.nv), lift_mask(lift_fix(xq[2], cur.shd), cur.nv)};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = (tid >> 5) + 8 * u;
                const float4 h = lb_vh(sVH[m], q);
                *reinterpret_cast<float4*>(sA + m * LB_TS + q4) = make_float4(uno_gelu(h.x), uno_gelu(h.y), uno_gelu(h.z), uno_gelu(h.w));
            }
        }
        if (tile + 2 < t_end) load_x(tile + 2, xq, qd);
        LB_STAMP(0);
        __syncthreads();
        LB_STAMP(1);
        f32x4 acc1[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc1[mt] = f32x4{0, 0, 0, 0};
        float wv[LB_CM / 4];
#pragma unroll
        for (int ks = 0; ks < LB_CM / 4; ++ks) wv[ks] = sW[(4 * ks + kk) * LB_WS + 16 * wave + r16];
#pragma unroll
        for (int ks = 0; ks < LB_CM / 4; ++ks) {
            const float* arow = sA + (4 * ks + kk) * LB_TS + r16;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc1[mt] = mfma16(arow[16 * mt], wv[ks], acc1[mt]);
        }
        LB_STAMP(2);
        float* const orow = sO + (16 * wave + r16) * LB_TS + 4 * kk;
        auto out_tile = [&](int mt) {
            *reinterpret_cast<float4*>(orow + 16 * mt) = make_float4(uno_gelu(acc1[mt][0] + b0v), uno_gelu(acc1[mt][1] + b0v),
                                                                     uno_gelu(acc1[mt][2] + b0v), uno_gelu(acc1[mt][3] + b0v));
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < LB_CM / 4; ++ks) {
            const float* arow = sA + (4 * ks + kk) * LB_TS + r16;
#pragma unroll
            for (int mt = 4; mt < 8; ++mt) acc1[mt] = mfma16(arow[16 * mt], wv[ks], acc1[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) out_tile(mt);
        __builtin_amdgcn_sched_group_barrier(0x100, 32, 0);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 4; mt < 8; ++mt) out_tile(mt);
        LB_STAMP(3);
        __syncthreads();
        LB_STAMP(1);
        // row-wise stores: thread -> (row (tid >> 5) + 8 u, quad tid & 31); slots past the domain's width are the padding: zeros
        if (cur.row_ok) {
            const int q4 = (tid & 31) * 4;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int o = (tid >> 5) + 8 * u;
                const float4 v = lift_mask(*reinterpret_cast<const float4*>(sO + o * LB_TS + q4), cur.nv);
                float* dst = ab + (size_t)o * G.Pp + cur.op;
                io_store4(dst, v.x, v.y, v.z, v.w);
                if (cur.last) {                         // the row's last columns (fewer than four)
                    if (p.tail > 0) dst[4] = 0.f;
                    if (p.tail > 1) dst[5] = 0.f;
                    if (p.tail > 2) dst[6] = 0.f;
                }
            }
        }
        LB_STAMP(7);
        __syncthreads();                // (the stores' LDS reads are done before the next tile's epilogue overwrites sO ... and sA)
        LB_STAMP(1);
    };
    for (int tile = t_begin; tile < t_end; tile += 2) {
        do_tile(tile, xqA, qdA);
        if (tile + 1 < t_end) do_tile(tile + 1, xqB, qdB);
    }
#ifdef UNO_LB_DEV
    if (p.stamps && lane == 0) {
        unsigned long long* o_ = p.stamps + (((size_t)b * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) o_[i] = tph[i];
    }
#endif
}

int launch_lift_forward_fused(const float* x, const float* w1, const float* b1, const float* w0, const float* b0, float* act, int B, int Cin,
                              int H, int W, int Hp, int Wp, hipStream_t s) {
    LiftFwdParams p;
    p.x = x; p.w1 = w1; p.b1 = b1; p.w0 = w0; p.b0 = b0; p.act = act;
    p.B = B; p.Cin = Cin; p.geo = lift_geom(H, W, Hp, Wp, Wp & ~3); p.tail = Wp & 3;
    p.stamps = nullptr;
    p.rev = next_sweep_reversed(SWEEP_LIFT);
#ifdef UNO_LB_DEV
    if (getenv("UNO_LF_STAMPS")) p.stamps = reinterpret_cast<unsigned long long*>((uintptr_t)strtoull(getenv("UNO_LF_STAMPS"), nullptr, 0));
#endif
    if ((long long)Hp * Wp * LB_CO >= (1LL << 31) || B > 65535 || (long long)H * Wp >= (1LL << 24)) { set_error("lift_forward: tensor too large"); return -2; }
    static int lds_slot[64];
    const size_t lds = sizeof(float) * (LB_CM * LB_TS + LB_CO * LB_TS + LB_CM * LB_WS) + sizeof(float4) * LB_CM;
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(lift_forward_kernel), lds, lds_slot)) { set_error("lift_forward: cannot raise dynamic LDS to %zu", lds); return -4; }
    {
        ProfScope prof("uno::lift_forward_kernel", 4.0 * B * ((double)H * W * Cin + (double)H * Wp * LB_CO), s);
        hipLaunchKernelGGL(lift_forward_kernel, dim3((unsigned)((p.geo.npt + LB_TPW - 1) / LB_TPW), (unsigned)B), dim3(256), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("lift_forward launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

static size_t lift_bwd_lds() { return sizeof(float) * (LB_CM * LB_TS + LB_CO * LB_TS + LB_CM * LB_WS + LB_CO * LB_WTS) + sizeof(float4) * LB_CM; }

bool lift_bwd_fused_applies(int Cin, int Cm, int Co, int W, long long P) {
    return Cin >= 1 && Cin <= 3 && Cm == LB_CM && Co == LB_CO && W >= 260 && P >= LB_PT && P < (1LL << 24) - (1 << 16);
}

long long lift_bwd_fused_parts(int B, int H, int W) {                // (64, 33) blocks of partial sums the fused kernel leaves
    const long long npt = ((long long)H * ((W + 3) & ~3) + LB_PT - 1) / LB_PT;
    return (long long)B * ((npt + LB_TPW - 1) / LB_TPW);
}

int launch_lift_backward_fused(const float* x, const float* w1, const float* b1, const float* w0, const float* b0, const float* g, float* part,
                               float* part1, int B, int Cin, int H, int W, int Hp, int Wp, hipStream_t s, const float* g2) {
    LiftBwdParams p;
    p.x = x; p.w1 = w1; p.b1 = b1; p.w0 = w0; p.b0 = b0; p.g = g; p.g2 = g2; p.part = part; p.part1 = part1;
    p.B = B; p.Cin = Cin; p.geo = lift_geom(H, W, Hp, Wp);
    p.stamps = nullptr;
    p.rev = next_sweep_reversed(SWEEP_LIFT);
#ifdef UNO_LB_DEV
    if (getenv("UNO_LB_STAMPS")) p.stamps = reinterpret_cast<unsigned long long*>((uintptr_t)strtoull(getenv("UNO_LB_STAMPS"), nullptr, 0));
#endif
    if ((long long)Hp * Wp * LB_CO >= (1LL << 31) || B > 65535) { set_error("lift_backward: tensor too large"); return -2; }
    static int lds_slot[64];
    const size_t lds = lift_bwd_lds();
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(lift_backward_kernel), lds, lds_slot)) { set_error("lift_backward: cannot raise dynamic LDS to %zu", lds); return -4; }
    {
        ProfScope prof("uno::lift_backward_kernel", 4.0 * B * (double)H * W * (Cin + LB_CO * (g2 ? 2 : 1)), s);
        hipLaunchKernelGGL(lift_backward_kernel, dim3((unsigned)((p.geo.npt + LB_TPW - 1) / LB_TPW), (unsigned)B), dim3(256), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("lift_backward launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
