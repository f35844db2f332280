// K1p / K3p - plane-batched variants of K1 / K3 for MANY SMALL images: the (dim2, dim3) planes of the 3-D layer
// (SpectralConv3d_Uno.forward, reference integral_operators.py:395-427: 16384 planes of 64 x 20 per block at config C4)
// and the coarse levels of the 2-D models (16 x 16, 32 x 32 at NS-2D).  K1 / K3 give every image a workgroup and pay
// their prologue (twiddle tables, tail tables, operand walk set-up) per image: at 5 KB per image that prologue is the
// run time (79-85 us for 84 MB = 1.4 TB/s).  Here a workgroup builds its tables once and its four waves then walk over
// images, one image per wave at a time.
//
// Both stages of both kernels run on v_mfma_f32_16x16x4_f32 with the complex values INTERLEAVED along a matrix
// dimension (column n = 2 l + c, c = 0 real / 1 imaginary part of mode l), which is also the memory layout of the
// truncated spectrum: 16 lanes read / write 64 contiguous bytes of a spectrum row.  A complex product needs the
// partner operand "swap (re, im) and negate one of them": one DPP quad-permute + one multiply per register.
//
//   K1p  T[h][n]  = sum_w x[h][w] G[w][n]            G[w][2l] = cos phi, G[w][2l+1] = -sin phi, phi = 2 pi l w / W
//        X[j][n]  = sum_h c(j,h) T[h][n] + s(j,h) T~[h][n]      T~[2l] = T[2l+1], T~[2l+1] = -T[2l];  theta = 2 pi K_j h / H
//        (x rows come from a per-wave LDS copy of the image, filled with 16-byte loads of the contiguous image; the
//        stage-A accumulators are directly the B operand of stage B: register r of lane group g is row 4 g + r)
//   K3p  U^T[n][h] = sum_j O'[j][n] c(j,h) + O~[j][n] s(j,h)     O~[2l] = -O'[2l+1], O~[2l+1] = O'[2l]
//        y^T[w][h] = sum_n G[w][n] U^T[n][h]                    -> every lane ends up with 4 consecutive columns of a row
// Same results as K1 / K3 to f32 rounding (plain instead of symmetric summation order).
#include "uno_common.h"
#include <algorithm>
#include <cstdio>
#include <mutex>
#include <vector>

namespace uno {

constexpr int PL_WAVES = 4;                 // waves per workgroup
constexpr int PL_MAX_ELEMS = 2048;          // largest image (floats) taken by K3p
constexpr int PL_FWD_MAX_ELEMS = 1792;      // ... by K1p (the next image waits in registers: 7 x 16 bytes per lane)
constexpr int PL_MIN_IMAGES = 128;          // below this K1 / K3 are as good (the chip is not filled either way)
constexpr size_t PL_MAX_LDS = 64 * 1024;
constexpr int PL_IMG_PAD = 8;                // zero floats behind a wave's image copy (stage A reads rows in 8-column steps)
constexpr int PL_PIECES = PL_FWD_MAX_ELEMS / 256;    // 16-byte pieces of an image per lane

__device__ __forceinline__ float lane_xor1(float v) {       // the value held by lane ^ 1 (DPP quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

struct PlaneShape {
    int nrt, kw, nwt, mt, ntn;
};
static PlaneShape plane_shape(const Dft2dParams& p) {
    return PlaneShape{(p.H + 15) / 16, (p.W + 3) / 4, (p.W + 15) / 16, (2 * p.m1 + 15) / 16, (2 * p.m2 + 15) / 16};
}
static size_t fwd_plane_lds(const Dft2dParams& p) {
    const PlaneShape g = plane_shape(p);
    const size_t img = (size_t)((p.H * p.W + 3) & ~3);
    return (size_t)g.mt * g.nrt * 4 * 64 * 8 + (size_t)(2 * ((p.W + 7) / 8)) * g.ntn * 64 * 4 + PL_WAVES * (img + PL_IMG_PAD) * 4;
}
static size_t inv_plane_lds(const Dft2dParams& p) {
    const PlaneShape g = plane_shape(p);
    return (size_t)g.nrt * 4 * g.mt * 64 * 8 + (size_t)g.nwt * g.ntn * 4 * 64 * 4;
}
static bool plane_shape_ok(const Dft2dParams& p) {
    if (p.bf16) return false;
    const long long hw = (long long)p.H * p.W;
    return (p.n_img >= PL_MIN_IMAGES || p.rowfreq) && hw >= 16 && hw <= PL_MAX_ELEMS && p.W <= 64 && 2 * p.m1 <= 48 && 2 * p.m2 <= 32 &&
           p.m1 >= 1 && p.m2 >= 1;
}
bool dft2d_fwd_plane_applies(const Dft2dParams& p) { return plane_shape_ok(p) && p.H * p.W <= PL_FWD_MAX_ELEMS && fwd_plane_lds(p) <= PL_MAX_LDS; }
bool dft2d_inv_plane_applies(const Dft2dParams& p) { return plane_shape_ok(p) && inv_plane_lds(p) <= PL_MAX_LDS; }

// ---------------------------------------------------------------------------------------------------------------- K1p
template <int MT, int NTN>
__global__ __launch_bounds__(64 * PL_WAVES, 2) void dft2d_fwd_plane_kernel(Dft2dParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int HW = H * W;
    const int nrt = (H + 15) >> 4, kw2 = (W + 7) >> 3;                     // stage A runs in pairs of k-steps (8 columns)
    const int img_stride = ((HW + 3) & ~3) + PL_IMG_PAD;
    float2* sTwB = reinterpret_cast<float2*>(smem);                         // [MT][nrt * 4][64]: (cos, sin) theta(j, h), stage-B A operand
    float* sTwA = reinterpret_cast<float*>(sTwB + MT * nrt * 4 * 64);       // [2 kw2][NTN][64]: G[w][n], stage-A B operand; zero for w >= W
    float* sImgAll = sTwA + 2 * kw2 * NTN * 64;                             // [PL_WAVES][img_stride]; the PL_IMG_PAD floats after an image stay zero
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int e = tid; e < MT * nrt * 4 * 64; e += 64 * PL_WAVES) {
        const int ln = e & 63, step = (e >> 6) % (nrt * 4), mt = (e >> 6) / (nrt * 4);
        const int j = 16 * mt + (ln & 15), h = 16 * (step >> 2) + 4 * (ln >> 4) + (step & 3);
        float2 v = make_float2(0.f, 0.f);
        if (j < 2 * m1 && h < H) v = p.twH[(unsigned)((p.rowfreq ? p.rowfreq[j] : corner_freq(j, m1, H)) * h) % (unsigned)H];
        sTwB[e] = v;
    }
    for (int e = tid; e < 2 * kw2 * NTN * 64; e += 64 * PL_WAVES) {
        const int ln = e & 63, tn = (e >> 6) % NTN, s = (e >> 6) / NTN;
        const int n = 16 * tn + (ln & 15), l = n >> 1, w = 4 * s + (ln >> 4);
        float v = 0.f;
        if (l < m2 && w < W) {
            const float2 t = p.twW[(unsigned)(l * w) % (unsigned)W];
            v = (n & 1) ? -t.y : t.x;
        }
        sTwA[e] = v;
    }
    for (int e = tid; e < PL_WAVES * PL_IMG_PAD; e += 64 * PL_WAVES)
        sImgAll[(size_t)(e / PL_IMG_PAD) * img_stride + ((HW + 3) & ~3) + e % PL_IMG_PAD] = 0.f;
    if ((HW & 3) && tid < PL_WAVES) {
        for (int e = HW; e < ((HW + 3) & ~3); ++e) sImgAll[(size_t)tid * img_stride + e] = 0.f;
    }
    __syncthreads();

    float* sImg = sImgAll + (size_t)wave * img_stride;
    const float* twA = sTwA + lane;
    const float2* twB = sTwB + lane;
    float cs[NTN];
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn) {
        const int l = (16 * tn + r16) >> 1;
        cs[tn] = l < m2 ? p.scale * (p.herm ? herm_weight(l, W) : 1.0f) : 0.f;
    }

    // The image (contiguous, 4-byte aligned) is fetched in 16-byte pieces, <= PL_PIECES per lane, one image AHEAD: the loads
    // of image i+1 are issued before image i is transformed and are written to the wave's LDS buffer when image i is done.
    // Piece q starts at float min(4 q, HW - 4): the last one is pulled back inside the image instead of running past it.
    const int nq = (HW + 3) >> 2;
    const int pieces = (nq + 63) >> 6;
    f4u pre[PL_PIECES];
    auto fetch = [&](int img) {
        const float* src = p.in + (size_t)min(img, p.n_img - 1) * HW;
#pragma unroll
        for (int i = 0; i < PL_PIECES; ++i)
            if (i < pieces) pre[i] = *reinterpret_cast<const f4u*>(src + min(4 * min(lane + 64 * i, nq - 1), HW - 4));
    };
    const int img0 = blockIdx.x * PL_WAVES + wave, img_step = gridDim.x * PL_WAVES;
    // LDS executes a wave's instructions in order, so the wave needs no barrier around its private buffer - only the
    // compiler must not reorder across the hand-over points
    auto stage = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // this image's reads of the buffer stay above
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < PL_PIECES; ++i) {
            const int q = lane + 64 * i;
            if (i < pieces && q < nq) {
                const int st = min(4 * q, HW - 4);
                if ((st & 3) == 0) {
                    *reinterpret_cast<f32x4*>(sImg + st) = f32x4{pre[i].v[0], pre[i].v[1], pre[i].v[2], pre[i].v[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sImg[st + e] = pre[i].v[e];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Order inside an iteration: transform image i (from LDS) -> copy image i+1 (fetched one iteration ago) to LDS -> issue
    // the loads of image i+2 -> store the spectrum of image i.  The one s_waitcnt vmcnt(0) (before the LDS copy) then only
    // sees loads and stores that were issued a whole transform earlier; with the stores first it waited for them every image.
    fetch(img0);
    stage();
    fetch(img0 + img_step);
    for (int img = img0; img < p.n_img; img += img_step) {
        f32x4 X[MT][NTN], X2[MT][NTN];          // cos / sin products accumulate separately (two MFMA chains per tile)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) { X[mt][tn] = f32x4{0, 0, 0, 0}; X2[mt][tn] = f32x4{0, 0, 0, 0}; }

        const int nrt_run = nrt;
        for (int t = 0; t < nrt_run; ++t) {
            // rows past H: finite data of the last row, zero twiddle in stage B; columns past W: the next row's (finite) data
            // or the zero pad behind the image, zero twiddle in the table
            const float* arow = sImg + min(16 * t + r16, H - 1) * W + kk;
            f32x4 T[NTN], Tb[NTN];          // even / odd k-steps: half the dependent-MFMA chain length
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) { T[tn] = f32x4{0, 0, 0, 0}; Tb[tn] = f32x4{0, 0, 0, 0}; }
#pragma unroll 2
            for (int s2 = 0; s2 < kw2; ++s2) {
                const float a0 = arow[8 * s2], a1 = arow[8 * s2 + 4];
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn) {
                    T[tn] = mfma16(a0, twA[((2 * s2) * NTN + tn) * 64], T[tn]);
                    Tb[tn] = mfma16(a1, twA[((2 * s2 + 1) * NTN + tn) * 64], Tb[tn]);
                }
            }
            f32x4 T2[NTN];
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) {
                T[tn] += Tb[tn];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float o = lane_xor1(T[tn][r]);
                    T2[tn][r] = (lane & 1) ? -o : o;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float2 tw = twB[((mt * nrt + t) * 4 + r) * 64];
#pragma unroll
                    for (int tn = 0; tn < NTN; ++tn) {
                        X[mt][tn] = mfma16(tw.x, T[tn][r], X[mt][tn]);
                        X2[mt][tn] = mfma16(tw.y, T2[tn][r], X2[mt][tn]);
                    }
                }
        }

        if (img + img_step < p.n_img) {
            stage();
            fetch(img + 2 * img_step);
        }
        float* out = p.out + spectrum_index(p, img) * (size_t)(4 * m1 * m2);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * mt + 4 * kk + r;
                if (j < 2 * m1) {
                    const float rf = (p.mask && !row_survives(j, m1, H)) ? 0.f : 1.f;
#pragma unroll
                    for (int tn = 0; tn < NTN; ++tn) {
                        const int n = 16 * tn + r16;
                        if (n < 2 * m2) out[(size_t)j * 2 * m2 + n] = (X[mt][tn][r] + X2[mt][tn][r]) * (cs[tn] * rf);
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------- K3p
// ACC: out += result, and act_out = gelu(out) where the caller asked for it (a separate instantiation: the plain stores stay as they are)
template <int MT, int NTN, bool ACC>
__global__ __launch_bounds__(64 * PL_WAVES, 2) void dft2d_inv_plane_kernel(Dft2dParams p) {
    constexpr int KSJ = 4 * MT;                                          // k-steps over the corner rows (compiled bound)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int nrt = (H + 15) >> 4, nwt = (W + 15) >> 4;
    const int ksj = (2 * m1 + 3) >> 2;                                   // k-steps actually needed
    float2* sTwB = reinterpret_cast<float2*>(smem);                      // [nrt][KSJ][64]: (cos, sin) theta(j, h), stage-B' B operand
    float* sTwA = reinterpret_cast<float*>(sTwB + nrt * KSJ * 64);       // [nwt][NTN][4][64]: G[w][n], stage-A' A operand
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int e = tid; e < nrt * KSJ * 64; e += 64 * PL_WAVES) {
        const int ln = e & 63, ks = (e >> 6) % KSJ, t = (e >> 6) / KSJ;
        const int h = 16 * t + (ln & 15), j = 4 * ks + (ln >> 4);
        float2 v = make_float2(0.f, 0.f);
        if (j < 2 * m1 && h < H) v = p.twH[(unsigned)((p.rowfreq ? p.rowfreq[j] : corner_freq(j, m1, H)) * h) % (unsigned)H];
        sTwB[e] = v;
    }
    for (int e = tid; e < nwt * NTN * 4 * 64; e += 64 * PL_WAVES) {
        const int ln = e & 63, r = (e >> 6) & 3, tn = (e >> 8) % NTN, wt = (e >> 8) / NTN;
        const int w = 16 * wt + (ln & 15), n = 16 * tn + 4 * (ln >> 4) + r, l = n >> 1;
        float v = 0.f;
        if (l < m2 && w < W) {
            const float2 t = p.twW[(unsigned)(l * w) % (unsigned)W];
            v = (n & 1) ? -t.y : t.x;
        }
        sTwA[e] = v;
    }
    __syncthreads();

    float cs[NTN];
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn) {
        const int l = (16 * tn + r16) >> 1;
        cs[tn] = l < m2 ? p.scale * (p.herm ? herm_weight(l, W) : 1.0f) : 0.f;
    }

    // The spectrum of image i+1 is fetched (raw, clamped addresses) while image i is transformed; scale, Hermitian weight
    // and the later-wins row mask are applied when it becomes the current operand.
    float Ov[KSJ][NTN], O2[KSJ][NTN], On[KSJ][NTN];
    auto fetch = [&](int img) {
        const float* O = p.in + spectrum_index(p, min(img, p.n_img - 1)) * (size_t)(4 * m1 * m2);
#pragma unroll
        for (int ks = 0; ks < KSJ; ++ks)
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn)
                if (ks < ksj) On[ks][tn] = O[min(4 * ks + kk, 2 * m1 - 1) * 2 * m2 + min(16 * tn + r16, 2 * m2 - 1)];
    };
    const int img0 = blockIdx.x * PL_WAVES + wave, img_step = gridDim.x * PL_WAVES;
    auto adopt = [&]() {           // fetched spectrum -> current operand (and its "swap, negate" partner)
#pragma unroll
        for (int ks = 0; ks < KSJ; ++ks) {
            const int j = 4 * ks + kk;
            const bool keep = j < 2 * m1 && !(p.mask && !row_survives(j, m1, H));
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) {
                Ov[ks][tn] = (ks < ksj && keep && 16 * tn + r16 < 2 * m2) ? On[ks][tn] * cs[tn] : 0.f;
                const float o = lane_xor1(Ov[ks][tn]);
                O2[ks][tn] = (lane & 1) ? o : -o;
            }
        }
    };
    // The hand-over to the next image (adopt + issue the loads of the one after) sits inside the LAST row tile, between its
    // column stage (the last use of the operand) and its row stage + stores: the s_waitcnt vmcnt(0) in front of adopt()
    // then does not wait for stores that were issued a moment ago.
    fetch(img0);
    adopt();
    fetch(img0 + img_step);
    for (int img = img0; img < p.n_img; img += img_step) {
        float* dst = p.out + (size_t)img * H * W;
        for (int t = 0; t < nrt; ++t) {
            // two accumulators per output tile (even / odd k-steps): half the dependent-MFMA chain length
            f32x4 U[NTN], Ub[NTN];
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) { U[tn] = f32x4{0, 0, 0, 0}; Ub[tn] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int ks = 0; ks < KSJ; ++ks) {          // k-steps past ceil(2 m1 / 4) multiply zeros: cheaper than branching around them
                const float2 tw = sTwB[(t * KSJ + ks) * 64 + lane];
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn) {
                    U[tn] = mfma16(Ov[ks][tn], tw.x, U[tn]);
                    Ub[tn] = mfma16(O2[ks][tn], tw.y, Ub[tn]);
                }
            }
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) U[tn] += Ub[tn];
            if (t == nrt - 1 && img + img_step < p.n_img) {
                __builtin_amdgcn_sched_barrier(0);
                adopt();
                fetch(img + 2 * img_step);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int h = 16 * t + r16;
            auto put = [&](const f32x4& Y, int wt) {
                const int w0 = 16 * wt + 4 * kk;
                if (h < H) {
                    float* row = dst + (size_t)h * W;
                    if constexpr (ACC) {
                        float* arow = p.act_out ? p.act_out + (size_t)img * H * W + (size_t)h * W : nullptr;
                        if (w0 + 3 < W) {
                            const f4u o = *reinterpret_cast<const f4u*>(row + w0);
                            const f4u v = f4u{{o.v[0] + Y[0], o.v[1] + Y[1], o.v[2] + Y[2], o.v[3] + Y[3]}};
                            *reinterpret_cast<f4u*>(row + w0) = v;
                            if (arow) *reinterpret_cast<f4u*>(arow + w0) = f4u{{uno_gelu(v.v[0]), uno_gelu(v.v[1]), uno_gelu(v.v[2]), uno_gelu(v.v[3])}};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (w0 + e < W) {
                                    const float v = row[w0 + e] + Y[e];
                                    row[w0 + e] = v;
                                    if (arow) arow[w0 + e] = uno_gelu(v);
                                }
                        }
                    } else if (w0 + 3 < W) {
                        *reinterpret_cast<f4u*>(row + w0) = f4u{{Y[0], Y[1], Y[2], Y[3]}};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (w0 + e < W) row[w0 + e] = Y[e];
                    }
                }
            };
            const int nwt_run = nwt;
            int wt = 0;
            for (; wt + 2 <= nwt_run; wt += 2) {          // two column tiles at a time: two independent MFMA chains
                f32x4 Y0 = f32x4{0, 0, 0, 0}, Y1 = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        Y0 = mfma16(sTwA[((wt * NTN + tn) * 4 + r) * 64 + lane], U[tn][r], Y0);
                        Y1 = mfma16(sTwA[(((wt + 1) * NTN + tn) * 4 + r) * 64 + lane], U[tn][r], Y1);
                    }
                put(Y0, wt);
                put(Y1, wt + 1);
            }
            if (wt < nwt_run) {
                f32x4 Y0 = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Y0 = mfma16(sTwA[((wt * NTN + tn) * 4 + r) * 64 + lane], U[tn][r], Y0);
                put(Y0, wt);
            }
        }
    }
}

// One wave of workgroups: as many as are resident at once (LDS / register bound), so that every wave walks over the same
// number of images (+-1) instead of a second, partly filled round of workgroups.
static int plane_grid(const void* kernel, size_t lds, int n_img) {
    // the occupancy query costs microseconds of host time: remember it per (device, kernel, LDS size)
    struct Key { int dev; const void* k; size_t lds; };
    struct Entry { Key key; int resident; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int resident = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Entry& e : cache)
            if (e.key.dev == dev && e.key.k == kernel && e.key.lds == lds) { resident = e.resident; break; }
        if (!resident) {
            int cus = 256, per_cu = 0;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64 * PL_WAVES, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            resident = cus * per_cu;
            cache.push_back(Entry{Key{dev, kernel, lds}, resident});
        }
    }
    const long long want = ((long long)n_img + PL_WAVES - 1) / PL_WAVES;
    // (resident = device CUs x workgroups per CU, cached; CUs set aside for communication kernels leave the grid proportionally smaller)
    const int r = reserved_cus();
    const long long mine = r > 0 ? std::max<long long>(1, (long long)resident * std::max(8, 256 - r) / 256) : resident;
    return (int)std::min<long long>(want, mine);
}

template <int MT, int NTN>
static int launch_fwd_plane_t(const Dft2dParams& p, hipStream_t s) {
    const size_t lds = fwd_plane_lds(p);
    const int grid = plane_grid(reinterpret_cast<const void*>(dft2d_fwd_plane_kernel<MT, NTN>), lds, p.n_img);
    {
        ProfScope prof("uno::dft2d_fwd_plane_kernel", (double)p.n_img * ((double)p.H * p.W * 4.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL((dft2d_fwd_plane_kernel<MT, NTN>), dim3(grid), dim3(64 * PL_WAVES), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_fwd_plane launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

template <int MT, int NTN>
static int launch_inv_plane_t(const Dft2dParams& p, hipStream_t s) {
    const size_t lds = inv_plane_lds(p);
    const void* kern = p.accumulate ? reinterpret_cast<const void*>(dft2d_inv_plane_kernel<MT, NTN, true>)
                                    : reinterpret_cast<const void*>(dft2d_inv_plane_kernel<MT, NTN, false>);
    const int grid = plane_grid(kern, lds, p.n_img);
    {
        const double img = (double)p.n_img * (double)p.H * p.W * 4.0;
        ProfScope prof(p.accumulate ? "uno::dft2d_inv_plane_kernel<acc>" : "uno::dft2d_inv_plane_kernel",
                       img * (p.accumulate ? (p.act_out ? 3.0 : 2.0) : 1.0) + (double)p.n_img * 2.0 * p.m1 * p.m2 * 8.0, s);
        if (p.accumulate) hipLaunchKernelGGL((dft2d_inv_plane_kernel<MT, NTN, true>), dim3(grid), dim3(64 * PL_WAVES), lds, s, p);
        else hipLaunchKernelGGL((dft2d_inv_plane_kernel<MT, NTN, false>), dim3(grid), dim3(64 * PL_WAVES), lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_inv_plane launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

#define UNO_PLANE_DISPATCH(fn)                                                            \
    const PlaneShape g = plane_shape(p);                                                  \
    if (g.mt == 1 && g.ntn == 1) return fn<1, 1>(p, s);                                   \
    if (g.mt == 2 && g.ntn == 1) return fn<2, 1>(p, s);                                   \
    if (g.mt == 3 && g.ntn == 1) return fn<3, 1>(p, s);                                   \
    if (g.mt == 1 && g.ntn == 2) return fn<1, 2>(p, s);                                   \
    if (g.mt == 2 && g.ntn == 2) return fn<2, 2>(p, s);                                   \
    if (g.mt == 3 && g.ntn == 2) return fn<3, 2>(p, s);                                   \
    set_error("dft2d plane path: unsupported tile configuration (%d, %d)", g.mt, g.ntn);  \
    return -2;

int launch_dft2d_fwd_plane(const Dft2dParams& p, hipStream_t s) { UNO_PLANE_DISPATCH(launch_fwd_plane_t) }
int launch_dft2d_inv_plane(const Dft2dParams& p, hipStream_t s) { UNO_PLANE_DISPATCH(launch_inv_plane_t) }
#undef UNO_PLANE_DISPATCH

}  // namespace uno
