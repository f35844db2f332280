// K1, full-tile form - pruned forward 2-D DFT with the image tile staged in LDS.
//
// Same mathematics and the same MFMA structure as dft2d_fwd_kernel (dft2d_fwd_kernel.h: symmetric row stage on
// v_mfma_f32_16x16x4_f32 / 4x4x1_16b, stage-A accumulators consumed as the stage-B operand in place), but the two things that
// bound that kernel are gone (measured):
//   * its A operand came straight from global memory as 16 rows x 64-byte pieces at 4-byte alignment, two pieces per
//     instruction, three chunks (6 KB) in flight per wave: latency-bound at 3.4 TB/s.  Here a wave's 16 x W tile - ONE
//     contiguous run of the image - is copied into LDS by direct-to-LDS loads (buffer_load_dwordx4 ... lds: 1 KB of whole
//     128-byte lines per instruction, no VGPRs, the entire 27 KB tile in flight at once), placed at its memory offset modulo
//     128 bytes; the next tile is requested as soon as the row stage has read the current one, so the transfer runs under
//     the column stage.
//   * its B operand (twiddles) was gathered from a W-entry table by an integer-walked index (3 VALU + a conflict-prone
//     gather per operand).  Here the workgroup tabulates cos / sin(2 pi l w / W) once in MFMA operand layout
//     [k-step][stream][lane]; the inner loop reads it at immediate offsets.
// 27 KB of tile per wave + 36 KB of table let four waves share a CU (one per SIMD) - enough for the MFMA pipe because nothing
// in a wave's instruction stream waits on HBM any more except the one s_waitcnt per tile.
#pragma once
#include "uno_common.h"
#include <algorithm>
#include <cstdio>

namespace uno {

constexpr int FT_TAILMAX = 5;           // tail <= 15 pairs + w = 0 + Nyquist column = 17 elements = 5 k-steps
constexpr size_t FT_LDS_BUDGET = 160 * 1024 - 2048;
constexpr int FT_AUX = 2;               // cache policy of the tile loads: 2 = non-temporal (each line is read once)
constexpr int FT_MAXW = 160;            // widest image the full-tile form takes (see fwd_ft_geometry)

__device__ __forceinline__ f32x4 ft_mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);      // lane layout: dft2d_fwd_kernel.h
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int NT, int MT, int R4>
__global__ __launch_bounds__(256) void dft2d_fwd_ft_kernel(Dft2dParams p) {
    constexpr int NTF = R4 > 0 ? NT - 1 : NT;       // full 16-mode streams
    constexpr int NQ = R4 > 0 ? R4 : 1;
    constexpr int NTFA = NTF > 0 ? NTF : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, W = p.W, m1 = p.m1, m2 = p.m2;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int NWT = nthreads >> 6;
    const int NW = p.nw;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const unsigned H8 = 8u * H;

    // column-pair bookkeeping: pairs (w, W-w), w = 1..P; singles w = 0 and (W even) w = W/2
    const int P = (W - 1) >> 1;
    const int nfull = P >> 4;                   // chunks of 16 pairs = 4 k-steps
    const int prem = P - (nfull << 4);
    const int ntail = prem + 1 + ((W & 1) ? 0 : 1);
    const int tailsteps = (ntail + 3) >> 2;
    const int nk = 4 * nfull + tailsteps;

    const int tile_stride = (16 * W + 32 + 3) & ~3;                            // tile + alignment phase
    float* sTile = reinterpret_cast<float*>(smem);                             // [NWT][tile_stride]
    float2* sTabF = reinterpret_cast<float2*>(sTile + (size_t)NWT * tile_stride);   // [nk][NTF][64]
    const int nka = 4 * nfull + FT_TAILMAX;                                    // allocated k-steps (prefetches run to the fifth tail step)
    float2* sTab4 = sTabF + (size_t)nka * NTF * 64;                            // [nk][R4][16]
    float2* sTwH = sTab4 + (size_t)nka * R4 * 16;
    int* sTailW = reinterpret_cast<int*>(sTwH + H);                            // [FT_TAILMAX][2][64]: left / right column of a tail element (-1 = none)

    const int slot = wave / NW, wsub = wave - slot * NW;
    const int image = sweep_x(p.rev) * (NWT / NW) + slot;
    const bool active = image < p.n_img;
    const int nrt = (H + 15) >> 4;
    float* buf = sTile + (size_t)wave * tile_stride;

    // ---- direct-to-LDS tile loads.  Buffer resource = [128-byte aligned start of the image, end of the tensor): offsets are
    // non-negative, anything past the tensor reads as zero.
    const float* timg = p.in + (size_t)(active ? image : 0) * H * W;
    const uintptr_t ibase = reinterpret_cast<uintptr_t>(timg) & ~uintptr_t(127);
    const int a0 = (int)((reinterpret_cast<uintptr_t>(timg) - ibase) >> 2);
    const unsigned long long span = reinterpret_cast<uintptr_t>(p.in + (size_t)p.n_img * H * W) - ibase;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(ibase), 0, (int)(unsigned)std::min<unsigned long long>(span, 0xffffffffull), 0x00020000);
    auto request_tile = [&](int rt) {
        // tile rt occupies floats [toff, toff + rows W) from the aligned base; LDS index i <-> float (toff & ~31) + i
        const int toff = a0 + rt * 16 * W;
        const int rows = min(16, H - 16 * rt);
        const int total = (toff & 31) + rows * W;
        const int npiece = (total + 255) >> 8;
        const unsigned v0 = (unsigned)(((toff & ~31) + 4 * lane) * 4);
        for (int i = 0; i < npiece; ++i)
            if (256 * i + 4 * lane < total)         // the last piece stops at the end of the tile (lanes beyond it are masked off)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(buf + 256 * i), 16, v0 + 1024u * (unsigned)i, 0, 0, FT_AUX);
    };
    if (active && wsub < nrt) request_tile(wsub);

    // ---- tables (built while the first tile is on its way)
    for (int n = tid; n < H; n += nthreads) sTwH[n] = p.twH[n];
    for (int e = tid; e < FT_TAILMAX * 64; e += nthreads) {
        const int ln = e & 63, sq = e >> 6;
        const int q = 4 * sq + (ln >> 4);
        const bool pair = q < prem;
        const bool nyq = (q == prem + 1) && !(W & 1);
        const int w = pair ? 1 + 16 * nfull + q : (nyq ? (W >> 1) : 0);
        sTailW[(sq * 2 + 0) * 64 + ln] = (pair || q == prem || nyq) ? w : -1;
        sTailW[(sq * 2 + 1) * 64 + ln] = pair ? W - w : -1;
    }
    for (int e = tid; e < nk * 64; e += nthreads) {
        const int ln = e & 63, q = e >> 6, ks = ln >> 4;
        unsigned w;
        if (q < 4 * nfull) {
            w = 1u + 16u * (q >> 2) + 4u * ks + (q & 3);
        } else {
            const int qt = 4 * (q - 4 * nfull) + ks;
            w = qt < prem ? 1u + 16u * nfull + qt : ((qt == prem + 1 && !(W & 1)) ? (unsigned)(W >> 1) : 0u);
        }
#pragma unroll
        for (int t = 0; t < NTF; ++t) {
            const unsigned l = (unsigned)min(16 * t + (ln & 15), m2 - 1);
            sTabF[((size_t)q * NTF + t) * 64 + ln] = p.twW[(w * l) % (unsigned)W];
        }
        if ((ln & 12) == 0) {
#pragma unroll
            for (int g = 0; g < R4; ++g) {
                const unsigned l = (unsigned)min(16 * NTF + 4 * g + (ln & 3), m2 - 1);
                sTab4[((size_t)q * R4 + g) * 16 + 4 * ks + (ln & 3)] = p.twW[(w * l) % (unsigned)W];
            }
        }
    }
    __syncthreads();

    // stage-B A operand rows (corner rows) owned by this lane
    int Kj[MT];
    bool jvalid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int j = 16 * mt + r16;
        jvalid[mt] = j < 2 * m1;
        Kj[mt] = jvalid[mt] ? corner_freq(j, m1, H) : 0;
    }
    f32x4 Xr[MT][NT], Xi[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) { Xr[mt][t] = f32x4{0, 0, 0, 0}; Xi[mt][t] = f32x4{0, 0, 0, 0}; }

    const float2* tabF = sTabF + lane;
    const float2* tab4 = sTab4 + 4 * kk + (lane & 3);

    if (active) {
        for (int rt = wsub; rt < nrt; rt += NW) {
            const int toff = a0 + rt * 16 * W;
            const int phase = toff & 31;
            const int rows = min(16, H - 16 * rt);
            const float* row = buf + phase + r16 * W;           // this lane's image row inside the LDS tile
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile has landed
            if (rows < 16) {
                // rows past the image: zero them (their products are masked in stage B, but 0 * garbage could be NaN)
                for (int i = phase + rows * W + lane; i < phase + 16 * W; i += 64) buf[i] = 0.f;
            }

            f32x4 Tr[NT], Tn[NT];           // Tn = -Im T
            f32x4 Qr[NQ], Qn[NQ];           // 4x4x1 accumulators of the 4-mode groups (R4 > 0)
#pragma unroll
            for (int t = 0; t < NT; ++t) { Tr[t] = f32x4{0, 0, 0, 0}; Tn[t] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int g = 0; g < NQ; ++g) { Qr[g] = f32x4{0, 0, 0, 0}; Qn[g] = f32x4{0, 0, 0, 0}; }
#define UNO_FT_MFMA(E_, D_, TWF_, TW4_)                                                   \
    do {                                                                                  \
        _Pragma("unroll") for (int t = 0; t < NTF; ++t) {                                 \
            Tr[t] = mfma16((E_), (TWF_)[t].x, Tr[t]);                                     \
            Tn[t] = mfma16((D_), (TWF_)[t].y, Tn[t]);                                     \
        }                                                                                 \
        _Pragma("unroll") for (int g = 0; g < R4; ++g) {                                  \
            Qr[g] = ft_mfma4((E_), (TW4_)[g].x, Qr[g]);                                   \
            Qn[g] = ft_mfma4((D_), (TW4_)[g].y, Qn[g]);                                   \
        }                                                                                 \
    } while (0)

            // ---- stage A, full chunks: lane (row r16, k-slot kk) owns column pairs w = 1 + 16 c + 4 kk + s, s = 0..3.
            // One wave per SIMD: image operands AND twiddles of chunk c + 1 are requested before the MFMAs of chunk c are issued
            // (sched_barrier keeps the requests there), so no LDS latency sits in front of an MFMA.
            // (two operand sets in ping-pong, as in the half-tile form: no copies of the next chunk's operands after every MFMA block)
            struct RowOps { float xl[4], xr[4]; float2 twF[4][NTFA], tw4[4][NQ]; };
            RowOps opA, opB;
            const float* pl = row + 1 + 4 * kk;                 // left columns of chunk 0
            const float* pr = row + W - 4 - 4 * kk;             // mirrored columns of chunk 0 (ascending address)
            auto load_ops = [&](RowOps& o, int c) {
                // k-steps 4 c .. 4 c + 3: chunk c, or (c = nfull) the first tail k-steps' twiddles with the last chunk's (unused) image values
                const int cn = min(c, max(nfull - 1, 0));
                const float2* tf = tabF + (size_t)(4 * c) * (NTF * 64);
                const float2* t4 = tab4 + (size_t)(4 * c) * (R4 * 16);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    o.xl[s] = pl[16 * cn + s]; o.xr[s] = pr[-16 * cn + s];
#pragma unroll
                    for (int t = 0; t < NTF; ++t) o.twF[s][t] = tf[(s * NTF + t) * 64];
#pragma unroll
                    for (int g = 0; g < R4; ++g) o.tw4[s][g] = t4[(s * R4 + g) * 16];
                }
            };
            auto chunk = [&](RowOps& cur, RowOps& nxt, int c) {
                load_ops(nxt, c + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float E = cur.xl[s] + cur.xr[3 - s];
                    const float D = cur.xl[s] - cur.xr[3 - s];
                    UNO_FT_MFMA(E, D, cur.twF[s], cur.tw4[s]);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            load_ops(opA, 0);
            {
                int c = 0;
                for (; c + 2 <= nfull; c += 2) { chunk(opA, opB, c); chunk(opB, opA, c + 1); }
                if (c < nfull) { chunk(opA, opB, c); opA = opB; }               // (odd chunk count: one copy per tile)
            }
            float2 (&twF)[4][NTFA] = opA.twF;
            float2 (&tw4)[4][NQ] = opA.tw4;
            // ---- tail k-steps: pairs beyond the last full chunk, then w = 0, then the Nyquist column; the twiddles of the first
            // four are already in twF / tw4
            {
                const float2* tf = tabF + (size_t)(4 * nfull) * (NTF * 64);
                const float2* t4 = tab4 + (size_t)(4 * nfull) * (R4 * 16);
                float TL[FT_TAILMAX], TR[FT_TAILMAX];
#pragma unroll
                for (int s = 0; s < FT_TAILMAX; ++s) {
                    const int wl = sTailW[(s * 2 + 0) * 64 + lane], wr = sTailW[(s * 2 + 1) * 64 + lane];
                    const float vl = row[max(wl, 0)], vr = row[max(wr, 0)];
                    TL[s] = wl >= 0 ? vl : 0.f; TR[s] = wr >= 0 ? vr : 0.f;
                }
                float2 twF5[NTFA], tw45[NQ];
#pragma unroll
                for (int t = 0; t < NTF; ++t) twF5[t] = tf[(4 * NTF + t) * 64];
#pragma unroll
                for (int g = 0; g < R4; ++g) tw45[g] = t4[(4 * R4 + g) * 16];
#pragma unroll
                for (int s = 0; s < FT_TAILMAX; ++s) {
                    if (s < tailsteps) {
                        const float E = TL[s] + TR[s];
                        const float D = TL[s] - TR[s];
                        if (s < 4) UNO_FT_MFMA(E, D, twF[s], tw4[s]);
                        else UNO_FT_MFMA(E, D, twF5, tw45);
                    }
                }
            }
#undef UNO_FT_MFMA
            // the row stage has read the tile: request the next one, it lands during stage B
            if (rt + NW < nrt) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                request_tile(rt + NW);
            }

            if constexpr (R4 > 0) {
                // 4x4x1 result: lane 16 ws + 4 rg + j, reg i = partial T[row 4 rg + i][mode 16 NTF + 4 g + j] of k-slot ws.
                // Sum over the four k-slots, then move to the 16x16x4 accumulator layout stage B consumes
                // (lane (kk, n), reg s = T[row 4 kk + s][mode n]); columns n >= 4 R4 of the last tile are zero.
                f32x4 lastR = f32x4{0, 0, 0, 0}, lastN = f32x4{0, 0, 0, 0};
                const int src = 20 * kk + (r16 & 3);
#pragma unroll
                for (int g = 0; g < R4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float vr = Qr[g][i], vn = Qn[g][i];
                        vr += __shfl_xor(vr, 16); vn += __shfl_xor(vn, 16);
                        vr += __shfl_xor(vr, 32); vn += __shfl_xor(vn, 32);
                        const float gr = __shfl(vr, src), gn = __shfl(vn, src);
                        if ((r16 >> 2) == g) { lastR[i] = gr; lastN[i] = gn; }
                    }
                Tr[NT - 1] = lastR;
                Tn[NT - 1] = lastN;
            }

            // ---- stage B: X[j][l] += exp(-i theta(j,h)) * T[h][l], h = 16 rt + 4 kk + s
            unsigned idxB[MT];
            float2 twB[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned i0 = 8u * (((unsigned)Kj[mt] * (unsigned)(16 * rt + 4 * kk)) % (unsigned)H);
                twB[mt] = lds_tw(sTwH, i0);
                idxB[mt] = wrap_add(i0, 8u * (unsigned)Kj[mt], H8);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool hvalid = (16 * rt + 4 * kk + s) < H;
                float2 twBn[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    twBn[mt] = lds_tw(sTwH, idxB[mt]);
                    idxB[mt] = wrap_add(idxB[mt], 8u * (unsigned)Kj[mt], H8);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bool v = hvalid && jvalid[mt];
                    const float ac = v ? twB[mt].x : 0.f;
                    const float ans = v ? -twB[mt].y : 0.f;
                    const float anc = -ac;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        Xr[mt][t] = mfma16(ac, Tr[t][s], Xr[mt][t]);
                        Xi[mt][t] = mfma16(anc, Tn[t][s], Xi[mt][t]);
                        Xr[mt][t] = mfma16(ans, Tn[t][s], Xr[mt][t]);
                        Xi[mt][t] = mfma16(ans, Tr[t][s], Xi[mt][t]);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) twB[mt] = twBn[mt];
            }
        }
    }

    // ---- several waves per image: deterministic tree reduction of the partial spectra through the (now free) tile buffers
    constexpr int NACC = MT * NT * 8;
    (void)NACC;
    if (NW > 1) __syncthreads();            // every wave of the workgroup is done with its tile buffer
    for (int stride = 2; stride >= 1; stride >>= 1) {
        if (stride >= NW) continue;
        if (wsub >= stride && wsub < 2 * stride) {
            float* dst = sTile + (size_t)(wave - stride) * tile_stride;       // the partner's buffer
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[((mt * NT + t) * 8 + r) * 64 + lane] = Xr[mt][t][r];
                        dst[((mt * NT + t) * 8 + 4 + r) * 64 + lane] = Xi[mt][t][r];
                    }
        }
        __syncthreads();
        if (wsub < stride && wsub + stride < NW) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        Xr[mt][t][r] += buf[((mt * NT + t) * 8 + r) * 64 + lane];
                        Xi[mt][t][r] += buf[((mt * NT + t) * 8 + 4 + r) * 64 + lane];
                    }
        }
        __syncthreads();
    }

    if (active && wsub == 0) {
        float2* out = reinterpret_cast<float2*>(p.out) + spectrum_index(p, image) * 2 * m1 * m2;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int l = 16 * t + r16;
            if (l >= m2) continue;
            const float cs = p.scale * (p.herm ? herm_weight(l, W) : 1.0f);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * mt + 4 * kk + r;
                    if (j < 2 * m1) {
                        const float f = (p.mask && !row_survives(j, m1, H)) ? 0.f : cs;
                        out[(size_t)j * m2 + l] = make_float2(Xr[mt][t][r] * f, Xi[mt][t][r] * f);
                    }
                }
        }
    }
}

// ---- launcher side
struct FwdFtGeometry { int nw, g; size_t lds; };

static int ft_device_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return usable_cus(cus);
}

static size_t fwd_ft_lds_bytes(const Dft2dParams& p, int NTF, int R4, int waves) {
    const int P = (p.W - 1) >> 1, nfull = P >> 4, prem = P - (nfull << 4);
    const int tailsteps = (prem + 1 + ((p.W & 1) ? 0 : 1) + 3) >> 2;
    const size_t nk = (size_t)4 * nfull + FT_TAILMAX;      // allocated k-steps
    (void)tailsteps;
    const size_t tile_stride = (size_t)((16 * p.W + 32 + 3) & ~3);
    const size_t red = (size_t)0;       // the reduction reuses the tile buffers
    return (size_t)waves * tile_stride * 4 + nk * ((size_t)NTF * 512 + (size_t)R4 * 128) + (size_t)p.H * 8 + FT_TAILMAX * 2 * 64 * 4 + red;
}

// NW in {1, 2, 4} waves per image, G = 4 / NW images per workgroup (fewer when the images do not fill the CUs)
static bool fwd_ft_geometry(const Dft2dParams& p, int NT, int MT, int R4, FwdFtGeometry* out) {
    const int NTF = R4 > 0 ? NT - 1 : NT;
    // the tile's rows sit W floats apart: W % 8 == 0 puts the 16 rows of an operand read on 4 or fewer LDS banks
    if (p.bf16 || p.rowfreq || p.W % 8 == 0 || ((p.W - 1) >> 1) < 16) return false;
    // Measured (tools/kbench.py, 1024 images x 64 ch): 111^2 18.5 us against 21.5 for the register-path kernel, 223^2 47.9 / 47.3,
    // 421^2 243 / 212: with one wave per SIMD the row stage's LDS / VALU work and its MFMAs run back to back instead of
    // overlapping (ablations: 64 us of operand traffic + 83 us of row-stage MFMAs + 40 us of column stage + 60 us of exposed
    // tile loads), which the three waves per SIMD of the register path hide.  Large tiles therefore stay on that kernel.
    if (p.W > FT_MAXW) return false;
    if ((size_t)MT * NT * 8 * 64 > (size_t)16 * p.W) return false;             // reduction slots must fit a tile buffer
    const int nrt = (p.H + 15) / 16, cus = ft_device_cu_count();
    long long best_cost = -1;
    for (int nw = 1; nw <= 4 && nw <= nrt; nw *= 2) {
        int g = 4 / nw;
        while (g > 1 && (long long)(p.n_img + g - 1) / g < cus) --g;
        while (g > 1 && fwd_ft_lds_bytes(p, NTF, R4, nw * g) > FT_LDS_BUDGET) --g;
        const size_t lds = fwd_ft_lds_bytes(p, NTF, R4, nw * g);
        if (lds > FT_LDS_BUDGET) continue;
        const long long per_cu = std::max<long long>(1, std::min<long long>((long long)(FT_LDS_BUDGET / lds), 16 / (nw * g)));
        if (per_cu * nw * g < 3 && (long long)p.n_img * nw >= 4LL * cus) continue;      // a CU should hold (nearly) one wave per SIMD
        const long long groups = (p.n_img + g - 1) / g;
        const long long rounds = (groups + cus * per_cu - 1) / (cus * per_cu);
        const long long cost = rounds * ((nrt + nw - 1) / nw);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; *out = FwdFtGeometry{nw, g, lds}; }
    }
    return best_cost >= 0;
}

template <int NT, int MT, int R4>
static int launch_fwd_ft(Dft2dParams p, const FwdFtGeometry& g, hipStream_t s) {
    auto k = dft2d_fwd_ft_kernel<NT, MT, R4>;
    static int lds_slot[64];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(k), g.lds, lds_slot)) { set_error("dft2d_fwd: cannot raise dynamic LDS to %zu", g.lds); return -4; }
    p.nw = g.nw;
    p.rev = next_sweep_reversed(SWEEP_K1);
    char name[64];
    snprintf(name, sizeof(name), "uno::dft2d_fwd_ft_kernel<%d, %d, %d>", NT, MT, R4);
    {
        ProfScope prof(name, (double)p.n_img * ((double)p.H * p.W * 4.0 + 2.0 * p.m1 * p.m2 * 8.0), s);
        hipLaunchKernelGGL(k, dim3((p.n_img + g.g - 1) / g.g), dim3(64 * g.nw * g.g), g.lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dft2d_fwd launch: %s", hipGetErrorString(e)); return -5; }
    return 0;
}

}  // namespace uno
