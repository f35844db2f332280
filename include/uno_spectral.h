/* uno_spectral.h - C ABI of the MI355X-native U-NO spectral-convolution hot path.
 *
 * Drop-in boundary for the path BASELINE.json's north_star names: the reference
 * (ashiq24/UNO) has no FFI - the path is Python calling torch ops - so each entry point
 * below replaces one span of reference Python; the binding a maintainer adds is the
 * ctypes stub shown in INTEGRATION.md (uno_amd/_native.py is that stub, in-tree).
 *
 * Conventions
 *  - every pointer is DEVICE memory of the current HIP device, borrowed for the call;
 *  - float tensors are contiguous float32; complex tensors are interleaved (re, im)
 *    float32 pairs (torch.complex64 / view_as_real layout), passed as float*;
 *  - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work;
 *  - return 0 on success, <0 on error (uno_last_error() gives the message; nothing was
 *    enqueued for argument errors).  Error classes mirror what the reference surfaces as
 *    torch RuntimeError: shape/mode incompatibilities.
 *  - truncated-spectrum layout: (batch, channels, corner-rows = 2*modes1, modes2) for 2-D
 *    with the "lo" corner rows first (rows [:modes1] of rfft2) then the "hi" corner rows
 *    (rows [-modes1:]).
 *  - no state is retained between calls except immutable per-(device, N) twiddle tables.
 */
#ifndef UNO_SPECTRAL_H
#define UNO_SPECTRAL_H

#ifdef __cplusplus
extern "C" {
#endif

#define UNO_SPECTRAL_ABI_VERSION 12

/* ABI version of the loaded library (== UNO_SPECTRAL_ABI_VERSION it was built with). */
int uno_abi_version(void);
/* ABI 12 (no reference counterpart).  `bytes` bytes of host data in a fresh device allocation of the current device that is never
 * freed (an operand table a binding caches per shape), complete on return; NULL on failure.  Unlike a plain hipMalloc + hipMemcpy it
 * may be called while a stream is being captured into a hipGraph - the library's own tables are built this way, so a shape that
 * first appears inside a capture does not end the capture. */
void* uno_upload_table(const void* host, long long bytes);

/* Message of the last failing call on this thread ("" if none). */
const char* uno_last_error(void);

/* Bytes of scratch `ws` the 2-D forward / backward entry points need. */
long long uno_spectral_conv2d_fwd_ws_bytes(int B, int Ci, int Co, int m1, int m2);
long long uno_spectral_conv2d_bwd_ws_bytes(int B, int Ci, int Co, int m1, int m2);

/* SpectralConv2d_Uno.forward - reference integral_operators.py:181-207
 *   x  (B, Ci, H, W) f32;  w1, w2 (Ci, Co, m1, m2) c64;  y (B, Co, Ho, Wo) f32 [out]
 *   xtrunc (B, Ci, 2*m1, m2) c64 [out] - the truncated rfft2(x, norm="forward"), the only
 *   tensor backward needs besides the weights.  ws: scratch of ..._fwd_ws_bytes. */
int uno_spectral_conv2d_forward(const float* x, const float* w1, const float* w2, float* y,
                                float* xtrunc, void* ws, int B, int Ci, int Co, int H, int W,
                                int Ho, int Wo, int m1, int m2, void* stream);

/* Autograd adjoint of the above (PyTorch's FftC2R/Bmm/CopySlices/FftR2C backward chain,
 * SURVEY.md section 3.3 / Appendix A.2).  Complex gradients follow PyTorch's convention
 * dL/dRe + i dL/dIm.  gx / (gw1, gw2) may be NULL to skip that gradient.
 *   gy (B, Co, Ho, Wo) f32;  xtrunc as saved by forward;  gx (B, Ci, H, W) f32 [out]
 *   gw1, gw2 (Ci, Co, m1, m2) c64 [out] */
int uno_spectral_conv2d_backward(const float* gy, const float* xtrunc, const float* w1,
                                 const float* w2, float* gx, float* gw1, float* gw2, void* ws,
                                 int B, int Ci, int Co, int H, int W, int Ho, int Wo, int m1,
                                 int m2, void* stream);

/* Mixed-precision form of the two entry points above (BASELINE.json config 5: bf16 activations, f32 accumulation; the
 * reference itself raises on bf16 input, so this is an opt-in extension of integral_operators.py:181-207, not a replacement):
 * x / y and gy / gx are bfloat16 (same shapes, contiguous), read and written as such by the pruned DFT kernels (8-byte
 * loads / stores of four values, << 16 on load, round-to-nearest-even on store); weights, the truncated spectrum, every
 * accumulation and the weight gradients stay f32 / c64.  Half-precision weight STORAGE is the caller's cast. */
int uno_spectral_conv2d_forward_bf16(const void* x, const float* w1, const float* w2, void* y,
                                     float* xtrunc, void* ws, int B, int Ci, int Co, int H, int W,
                                     int Ho, int Wo, int m1, int m2, void* stream);
int uno_spectral_conv2d_backward_bf16(const void* gy, const float* xtrunc, const float* w1,
                                      const float* w2, void* gx, float* gw1, float* gw2, void* ws,
                                      int B, int Ci, int Co, int H, int W, int Ho, int Wo, int m1,
                                      int m2, void* stream);

/* The FFT crop / resample of pointwise_op_3D (reference integral_operators.py:448-463: unnormalised rfftn, four corners of
 * half the OUTPUT size copied into an INPUT-sized zero spectrum, irfftn(s = output size)) as a pruned DFT - pruned inverse DFT
 * pair with explicit frequency tables: along a complex axis the reference keeps spectrum indices r_j and reads index r_j as
 * frequency r_j of the output-length transform (irfftn trims / zero-pads at the END of the axis - bug-compatible, including
 * the misplaced negative frequencies when sizes differ).
 *   x (n_vol, D1, D2, D3) f32 -> y (n_vol, M1, M2, M3) f32;  f1_in / f1_out (J1 ints, device), f2_in / f2_out (J2 ints, device):
 *   forward / inverse frequency of kept row j along axes 1 / 2 (J1, J2 even; J1 <= 80, J2 <= 48); m3 kept half-spectrum bins;
 *   herm_in / herm_out: Hermitian column weights on the forward / inverse side (0 / 1 for the operator, 1 / 0 for its adjoint,
 *   which is the same call with sizes and tables swapped).  Planes of 16 ... 1792 (in) / 2048 (out) elements, D3, M3 <= 64. */
long long uno_fft_resample3d_ws_bytes(int n_vol, int D1, int M1, int J1, int J2, int m3);
int uno_fft_resample3d(const float* x, float* y, void* ws, int n_vol, int D1, int D2, int D3, int M1, int M2, int M3,
                       int J1, const int* f1_in, const int* f1_out, int J2, const int* f2_in, const int* f2_out, int m3,
                       float scale, int herm_in, int herm_out, void* stream);

/* The same, ACCUMULATING into y (y += resampled x) and, with y_act != NULL, writing y_act = gelu(y) in the same pass (ABI 7): the
 * point-wise branch of OperatorBlock_3D (reference integral_operators.py:506-512: x1_out + x2_out, then F.gelu) lands in the buffer
 * the spectral branch wrote - neither the sum nor the activation is a separate pass. */
int uno_fft_resample3d_acc(const float* x, float* y, float* y_act, void* ws, int n_vol, int D1, int D2, int D3, int M1, int M2, int M3,
                           int J1, const int* f1_in, const int* f1_out, int J2, const int* f2_in, const int* f2_out, int m3,
                           float scale, int herm_in, int herm_out, void* stream);

/* SpectralConv3d_Uno.forward - reference integral_operators.py:385-427
 *   x (B, Ci, H, W, T) f32;  w[0..3] = weights1..4 (Ci, Co, m1, m2, m3) c64 in the reference's corner
 *   order (lo,lo), (hi,lo), (lo,hi), (hi,hi);  y (B, Co, Ho, Wo, To) f32 [out]
 *   xtrunc (B, Ci, 4, m1, m2, m3) c64 [out]: truncated rfftn(x, norm="forward"), corner-major. */
long long uno_spectral_conv3d_fwd_ws_bytes(int B, int Ci, int Co, int H, int Ho, int m1, int m2, int m3);
long long uno_spectral_conv3d_bwd_ws_bytes(int B, int Ci, int Co, int H, int Ho, int m1, int m2, int m3);
int uno_spectral_conv3d_forward(const float* x, const float* const* w, float* y, float* xtrunc, void* ws,
                                int B, int Ci, int Co, int H, int W, int T, int Ho, int Wo, int To,
                                int m1, int m2, int m3, void* stream);
/* Adjoint; gx or gw (array of 4 output pointers) may be NULL to skip that gradient. */
int uno_spectral_conv3d_backward(const float* gy, const float* xtrunc, const float* const* w, float* gx,
                                 float* const* gw, void* ws, int B, int Ci, int Co, int H, int W, int T,
                                 int Ho, int Wo, int To, int m1, int m2, int m3, void* stream);

/* Pruned complex DFT along the leading axis of a 3-D transform:
 *   inverse = 0: planes (n_img, H, 2*m2*m3) c64 -> corner-major spectrum (n_img, 4, m1, m2, m3) c64
 *   inverse = 1: the reverse direction (e^{+i...}); mask_overlap applies on the spectrum side. */
int uno_cdft_axis(const float* in, float* out, int inverse, int n_img, int H, int m1, int m2, int m3,
                  float scale, int mask_overlap, void* stream);

/* Stage-level entry points (the three kernels the two calls above are built from). */

/* Pruned forward DFT: spec[img][j][l] = scale * c_l * keep_j * sum x e^{-2 pi i (K_j h/H + l w/W)}
 *   images (n_img, H, W) f32 -> spec (n_img, 2*m1, m2) c64.
 *   hermitian_cols: multiply column l by 1 (l = 0, Nyquist) or 2 - used for gO in backward.
 *   mask_overlap:   zero lo-corner rows j >= H - m1 (later slice-assignment wins). */
int uno_dft2d_forward(const float* images, float* spec, int n_img, int H, int W, int m1, int m2,
                      float scale, int hermitian_cols, int mask_overlap, void* stream);

/* Pruned inverse DFT: images[h][w] = Re sum_{j,l} scale * c_l * keep_j * spec[j][l] e^{+2 pi i (...)}. */
int uno_dft2d_inverse(const float* spec, float* images, int n_img, int H, int W, int m1, int m2,
                      float scale, int hermitian_cols, int mask_overlap, void* stream);

/* ABI 11.  Data parallelism (reference train loops are single-process; BASELINE.json north_star: RCCL all-reduce beside the backward
 * pass): set aside `n` compute units for concurrently running communication kernels.  Launch geometries that size themselves to the
 * device ("one workgroup per CU", persistent grids) then count on the remaining CUs.  Returns the previous value; 0 = none (default). */
int uno_reserve_cus(int n);

/* ABI 11.  (no reference counterpart) Alternating sweep direction: consecutive launches of the streaming kernels walk their work items
 * (images, batch entries, pixel tiles) in opposite directions, so that a kernel starts with the part of a > 256 MB tensor its predecessor
 * touched last - the part the Infinity Cache still holds.  Results do not depend on it.  enable = 0 turns it off (every launch front to
 * back); returns the previous setting (default 1). */
int uno_sweep_alternation(int enable);

/* ABI 11.  Pruned inverse DFT PLUS the up-sampled point-wise branch in one pass over the output:
 *   images[h][w] = (uno_dft2d_inverse's result) + sum_{u,v} Rh[h][u] Rw[w][v] addend[u][v]
 * = `x1_out + x2_out` of an up-sampling operator block, reference integral_operators.py:272-273, with x2_out the bicubic /
 * align_corners / antialias interpolation (:240-242) of the LOW-resolution 1x1-convolution result `addend` (n_img, Hs, Ws) - the
 * convolution and the interpolation commute - and, with the transposed operators, the input gradient of a down-sampling block.
 * Rh (H x Hs) and Rw (W x Ws) are banded (<= 12 source rows per 16 output rows, <= 12 source columns per 16 output columns:
 * up-sampling by two or more) and given as operand tables the caller builds once per size pair (uno_amd/resample.py
 * upsample_add_tables):
 *   tile_p0 [ceil(H/16)]           first source row of each 16-row output tile
 *   row_op  [ceil(H/16)][3][64]    entry (tile, e, lane) = Rh[16 tile + lane % 16][tile_p0 + 3 (lane / 16) + e]
 *   col_v0  [nwt][2]               first source column of (column tile, side), nwt = ((W/2) + 16) / 16; <= Ws - 12
 *   col_op  [nwt][2][3][64]        entry (wt, side, ks, lane) = Rw[w][col_v0 + 3 (lane / 16) + ks], w = 16 wt + lane % 16 (side 0)
 *                                  or W - 16 wt - lane % 16 (side 1); zero for w outside [0, W)
 * uno_dft2d_inverse_add_applies: 1 where the fused kernel exists (float32, rows of 192..223 or 416..447 elements whose 16-row
 * tiles fit the LDS, modes1 <= 27, modes2 <= 24), else 0 - the caller then runs uno_dft2d_inverse followed by uno_resample2d. */
int uno_dft2d_inverse_add_applies(int n_img, int H, int W, int m1, int m2, int Hs, int Ws);
int uno_dft2d_inverse_add(const float* spec, float* images, int n_img, int H, int W, int m1, int m2, float scale, int hermitian_cols,
                          int mask_overlap, const float* addend, int Hs, int Ws, const int* tile_p0, const float* row_op,
                          const int* col_v0, const float* col_op, void* stream);

/* The two transforms with bfloat16 images (the stages of uno_spectral_conv2d_*_bf16); spectra are c64 as above. */
int uno_dft2d_forward_bf16(const void* images, float* spec, int n_img, int H, int W, int m1, int m2,
                           float scale, int hermitian_cols, int mask_overlap, void* stream);
int uno_dft2d_inverse_bf16(const float* spec, void* images, int n_img, int H, int W, int m1, int m2,
                           float scale, int hermitian_cols, int mask_overlap, void* stream);

/* The same transforms with the spectra of a (B, group) batch of images placed at channels [offset, offset + group) of a
 * (B, stride, 2*m1, m2) spectrum tensor (image i <-> spectrum (i / group) * stride + offset + i % group): lets an
 * operator block consume torch.cat([x1, x2], dim=1) (reference darcy_flow_uno2d.py:117-125) from its two sources. */
int uno_dft2d_forward_grouped(const float* images, float* spec, int n_img, int H, int W, int m1, int m2, float scale,
                              int hermitian_cols, int mask_overlap, int group, int stride, int offset, void* stream);
int uno_dft2d_inverse_grouped(const float* spec, float* images, int n_img, int H, int W, int m1, int m2, float scale,
                              int hermitian_cols, int mask_overlap, int group, int stride, int offset, void* stream);

/* Per-mode channel mixing on the truncated spectrum, `ncorner` weight tensors of
 * `modes_per_corner` modes each (2-D: ncorner = 2, modes_per_corner = m1*m2).
 *   op 0: out[b,o] = sum_i in[b,i] * w[i,o]        (einsum "bixy,ioxy->boxy", :178-179)
 *   op 1: out[b,i] = sum_o in[b,o] * conj(w[i,o])  (grad wrt the input spectrum)
 *   in (B, Cin, ncorner*modes) c64, w[c] (Ci, Co, modes) c64, out (B, Cout, ncorner*modes) c64 */
int uno_mode_mix(const float* in, const float* const* w, float* out, int op, int B, int Ci, int Co,
                 int ncorner, int modes_per_corner, void* stream);

/* gw[c][i,o] = sum_b conj(xtrunc[b,i]) * go[b,o] per mode. */
int uno_mode_wgrad(const float* xtrunc, const float* go, float* const* gw, int B, int Ci, int Co,
                   int ncorner, int modes_per_corner, void* stream);
/* ABI 11.  Both per-mode GEMMs of a backward pass - the autograd adjoints of the einsum at reference integral_operators.py:178-179 /
 * :382-383 - from one launch where the kernels allow (else two): gx_spec[b,i] = sum_o go[b,o] conj(w[i,o]) and
 * gw[i,o] (+)= sum_b conj(xtrunc[b,i]) go[b,o]; `accumulate` adds into gw. */
int uno_mode_backward(const float* xtrunc, const float* go, const float* const* w, float* gx_spec, float* const* gw, int B, int Ci,
                      int Co, int ncorner, int modes_per_corner, int accumulate, void* stream);
/* uno_mode_wgrad with accumulate != 0: gw[c] += (in-place accumulation into a parameter's gradient buffer). */
int uno_mode_wgrad_acc(const float* xtrunc, const float* go, float* const* gw, int B, int Ci, int Co, int ncorner,
                       int modes_per_corner, int accumulate, void* stream);
/* uno_spectral_conv2d_backward in all three storage formats (io_format 0: f32 images; 1: bf16 images; 2: bf16 images + fp16 (re, im)
 * weights) with accumulate_gw != 0: gw1 / gw2 += instead of =. */
int uno_spectral_conv2d_backward_acc(const void* gy, const float* xtrunc, const void* w1, const void* w2, void* gx,
                                     float* gw1, float* gw2, void* ws, int B, int Ci, int Co, int H, int W, int Ho, int Wo,
                                     int m1, int m2, int io_format, int accumulate_gw, void* stream);

/* Separable banded resampling out = A . in . B^T of n_img images (H, W) -> (Ho, Wo): the resampling half of
 * pointwise_op_2D (reference integral_operators.py:240-242, bicubic / align_corners / antialias) and, with the
 * transposed band tables, its adjoint.  Row i of A has its first nonzero at column startH[i] and KH weights
 * wtH[i*KH .. i*KH+KH-1] (zero padded); likewise B.  tmp: scratch of 4*n_img*min(Ho*W, H*Wo) bytes.
 * Optional dense row-tile form of A for the fused single-pass kernel (NULL / 0 to use the two-pass kernels):
 * tile k covers output rows 16k..16k+15, reads input rows tile_p0[k] .. tile_p0[k]+NP-1 and has weights
 * tile_w[(k*NP + u)*16 + r] = A[16k + r][tile_p0[k] + u].
 * accumulate != 0: out += A . in . B^T (the spectral and the point-wise branch of an operator block write one
 * buffer: reference integral_operators.py:273 `x1_out + x2_out`, and the sum of their input gradients). */
int uno_resample2d(const float* in, float* out, float* tmp, int n_img, int H, int W, int Ho, int Wo,
                   const int* startH, const float* wtH, int KH, const int* startW, const float* wtW, int KW,
                   const int* tile_p0, const float* tile_w, int NP, int accumulate, void* stream);

/* Channel mixing of a channels-first tensor, y[b][o][p] = sum_i Wm(o,i) x[b][i][p] (+ bias[o]): the 1x1
 * convolution of pointwise_op_2D / pointwise_op_3D (reference integral_operators.py:219, 439: nn.Conv2d/3d(in,
 * out, 1)) and the lift / projection nn.Linear layers of the U-NO models (navier_stokes_uno2d.py fc0/fc1/fc2)
 * applied without the channels-last permute.  x (B, Ci, P), y (B, Co, P), P = pixels per sample (contiguous).
 * transpose_w = 0: w is (Co, Ci) row-major; transpose_w = 1: w is (Ci, Co) row-major and Wm = w^T (this is
 * the input-gradient call: x := grad_y, Ci := forward Co).  bias may be NULL.  accumulate != 0: y += (as in
 * uno_resample2d).
 * Fused activation forms for a layer whose input tensor is kept PRE-activation (the model's `fc(F.gelu(t))` patterns,
 * reference darcy_flow_uno2d.py:98-101, 128-131): act_in != 0 applies the exact-erf GELU to x as it is read (y = Wm gelu(x)
 * + bias); dgelu_of != NULL (B, Co, P) multiplies the product by gelu'(dgelu_of) - the input-gradient call then returns the
 * gradient of the pre-activation tensor.  The two are mutually exclusive.  accumulate = 2 (with dgelu_of): y = (y + product) *
 * gelu'(dgelu_of) - the call that adds the LAST contribution to the gradient of a block's activation also applies the block's
 * GELU derivative to the completed sum (reference integral_operators.py:282-283 backward), no separate pass. */
int uno_channel_mix(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co,
                    long long P, int transpose_w, int accumulate, int act_in, const float* dgelu_of, void* stream);

/* Weight / bias gradient of uno_channel_mix: gw[o][i] = sum_{b,p} gy[b][o][p] x[b][i][p], gb[o] = sum gy[b][o][p]
 * (gb may be NULL).  ws: scratch of uno_channel_wgrad_ws_bytes() bytes; partial sums are combined in a fixed
 * order (bit-reproducible run to run).  act_x != 0: x := gelu(x) as it is read (see uno_channel_mix). */
long long uno_channel_wgrad_ws_bytes(int B, int Ci, int Co, long long P);
int uno_channel_wgrad(const float* gy, const float* x, float* gw, float* gb, void* ws, int B, int Ci, int Co,
                      long long P, int act_x, void* stream);

/* uno_channel_mix / uno_channel_wgrad for a layer whose input is the channel concatenation of TWO tensors (the skip
 * connections of the U-NO models: reference darcy_flow_uno2d.py:117-127 `torch.cat([x_c4, x_c0], dim=1)` -> conv5,
 * `torch.cat([x_c5, x_fc0], dim=1)` -> fc1) without building the concatenation, one pass over every operand:
 *   x1 (B, C1, P) holds input channels [0, C1), x2 (B, Ci - C1, P) the rest (x2 = NULL: one source, C1 ignored);
 *   y1 (B, Co1, P) receives output channels [0, Co1), y2 (B, Co - Co1, P) the rest (y2 = NULL: one destination) - the
 *   transposed call on grad_y then yields the two input gradients of the layer from ONE read of grad_y;
 *   w is the layer's full (Co, Ci) weight (transpose_w = 1: (Ci, Co)), bias (Co) or NULL;
 *   act_in applies to x1 only, dgelu_of (B, Co1, P) to y1 only (the pre-activation source of the pair);
 *   y_act (B, Co, P) or NULL: additionally receives gelu(y) - the activation of a block without normalisation written by the
 *   kernel that completes the pre-activation sum (reference integral_operators.py:282-283), single destination only;
 *   proj_w (Co), proj_b (1) or NULL, proj_out (B, P): the call also writes proj_out[b][p] = proj_b + sum_o proj_w[o] gelu(y[b][o][p]),
 *   the one-channel projection `fc2(F.gelu(fc1(x)))` that ends the models (darcy_flow_uno2d.py:128-131) in the pass that produces
 *   y (which is still written: the backward of uno_gelu_project_forward needs it); Co <= 64, single destination.
 * Splits must be multiples of 16 (C1) / 64 (Co1; 128 when Co is a multiple of 128) channels; uno_channel_wgrad2 needs
 * C1 % 64 == 0 and P >= 64 (x2 = NULL: any shape).  gw is the full (Co, Ci) gradient; accumulate != 0: gw / gb += (the kernels
 * write a parameter's gradient buffer in place: the unrolled roll-out of ns_train_2d.py:46-68 sums 40 contributions per weight).
 * uno_channel_wgrad2 with accumulate = 3 runs the first stage only: the split-K partial sums stay in ws
 * (uno_channel_wgrad_ws_bytes() bytes = nparts blocks of (Co, Ci + 1) floats, bias sums in column Ci), gw / gb are not touched
 * (may be NULL); uno_channel_wgrad_finish() then sums `nparts` CONSECUTIVE blocks - the ws of one call, or of the T calls of a
 * roll-out laid out one after the other - into gw / gb (gb may be NULL) in the fixed order of the blocks: one second stage per
 * layer and training step instead of one per use. */
int uno_channel_mix2(const float* x1, const float* x2, int C1, const float* w, const float* bias, float* y1, float* y2, int Co1,
                     float* y_act, int B, int Ci, int Co, long long P, int transpose_w, int accumulate, int act_in,
                     const float* dgelu_of, const float* proj_w, const float* proj_b, float* proj_out, void* stream);
int uno_channel_wgrad2(const float* gy, const float* x1, const float* x2, int C1, float* gw, float* gb, void* ws, int B, int Ci,
                       int Co, long long P, int act_x, int accumulate, void* stream);
int uno_channel_wgrad_finish(const void* parts, float* gw, float* gb, int Ci, int Co, long long nparts, int accumulate, void* stream);

/* Final projection of the U-NO models fused with the GELU in front of it (reference darcy_flow_uno2d.py:128-131:
 * `x_fc1 = F.gelu(self.fc1(x)); x_out = self.fc2(x_fc1)` with fc2 = Linear(C, 1)), channels-first:
 *   out[b][p] = bias[0] + sum_c w[c] * gelu(pre[b][c][p])          pre (B, C, P), out (B, P), exact-erf GELU, bias may be NULL
 * backward: gpre = gelu'(pre) * w[c] * gout,  gw[c] = sum gout * gelu(pre),  gb[0] = sum gout  (gb may be NULL);
 * ws: scratch of uno_gelu_project_bwd_ws_bytes() bytes (fixed-order partial sums). */
int uno_gelu_project_forward(const float* pre, const float* w, const float* bias, float* out, int B, int C, long long P,
                             void* stream);
long long uno_gelu_project_bwd_ws_bytes(int B, int C, long long P);
int uno_gelu_project_backward(const float* pre, const float* w, const float* gout, float* gpre, float* gw, float* gb,
                              void* ws, int B, int C, long long P, void* stream);

/* The same three calls on a WINDOW of a wider plane (ABI 10).  The reference crops the domain padding before its last two layers
 * (darcy_flow_uno2d.py:125-131: `x_c5[..., :-padding, :-padding]`, then fc1 - GELU - fc2 on S x S points); here those layers read
 * the padded (S + pad)^2 tensors in place and touch the domain only: the pixel axis of the call is rows x cols logical pixels,
 * row r starting r * pitch elements into a channel plane, channel planes `plane` elements apart (plane >= rows * pitch) - for
 * EVERY operand of the call (x1, x2, y1, y2, y_act, dgelu_of; gy; pre, gpre; proj_out and gout with one plane per batch entry).
 * cols is a multiple of 4 with 260 <= cols <= pitch and rows * cols < 2^24: pass the domain width rounded UP to a multiple of 4 -
 * the extra columns are ordinary points of the padded grid (their output gradient is zero).  Elements outside the window are
 * neither read nor written: a gradient tensor produced by a windowed call must have its border cleared by the caller.
 * Float32 only; uno_channel_mix2_win runs the tiled and split kernels (not the wide / few-input forms), uno_channel_wgrad2_win needs
 * more than 4 input channels; scratch sizes as for the dense calls with P = rows * cols. */
int uno_channel_mix2_win(const float* x1, const float* x2, int C1, const float* w, const float* bias, float* y1, float* y2, int Co1,
                         float* y_act, int B, int Ci, int Co, int rows, int cols, int pitch, long long plane, int transpose_w,
                         int accumulate, int act_in, const float* dgelu_of, const float* proj_w, const float* proj_b, float* proj_out,
                         void* stream);
int uno_channel_wgrad2_win(const float* gy, const float* x1, const float* x2, int C1, float* gw, float* gb, void* ws, int B, int Ci,
                           int Co, int rows, int cols, int pitch, long long plane, int act_x, int accumulate, void* stream);
int uno_gelu_project_backward_win(const float* pre, const float* w, const float* gout, float* gpre, float* gw, float* gb, void* ws,
                                  int B, int C, int rows, int cols, int pitch, long long plane, void* stream);
/* ABI 12.  The backward pass of the models' last two layers in two launches, the gradient at fc1's output never in memory
 * (reference darcy_flow_uno2d.py:125-131, `x = self.fc1(x); x = F.gelu(x); x = self.fc2(x)` on `torch.cat([x_c5, x_fc0], dim=1)`;
 * uno_gelu_project_backward wrote that gradient - 726 MB at 421^2, batch 16 - and the input-gradient and weight-gradient calls each
 * read it back).  With pre = fc1's output kept by the forward call, w2 = fc2's weights (Co) and gout (B, plane) the gradient at the
 * model output, gy[b][o][q] = w2[o] gelu'(pre[b][o][q]) gout[b][q] is formed where the two kernels stage their operand:
 *   g1 (B, C1, plane), g2 (B, Ci - C1, plane) = w^T gy, split at the sources (g1 multiplied by gelu'(x1) when act_in: x1 was read through
 *     the GELU); gw (Co, Ci) [+]= gy [gelu](x1) | x2 ^T, gb (Co) [+]= sum gy (accumulate_w = 1 adds; gb may be NULL);
 *     gw2 (Co) = sum gelu(pre) gout, gb2 (1) = sum gout (overwritten; gb2 may be NULL).
 * Geometry as the *_win calls (rows x cols window of planes `plane` elements apart, row r at r * pitch; elements outside the window are
 * neither read nor written), or rows = cols = pitch = 0: dense planes of `plane` pixels.  x2 = NULL: one source (C1 is ignored).
 * uno_project_backward_applies: 1 where both kernels take the shape (float32; Ci a multiple of 128, Co a multiple of 16 below 128, sources
 * split at a multiple of 64, >= 100 000 pixels in all, a multiple of 4 per plane) - elsewhere use the three separate calls.
 * ws: uno_project_backward_ws_bytes(B, Ci, Co, pixels per plane = rows * cols) bytes. */
int uno_project_backward_applies(int B, int C1, int Ci, int Co, int rows, int cols, int pitch, long long plane);
long long uno_project_backward_ws_bytes(int B, int Ci, int Co, long long P);
int uno_project_backward(const float* x1, const float* x2, int C1, const float* w, const float* pre, const float* w2, const float* gout,
                         float* g1, float* g2, float* gw, float* gb, float* gw2, float* gb2, void* ws, int B, int Ci, int Co, int rows,
                         int cols, int pitch, long long plane, int act_in, int accumulate_w, void* stream);
/* Everything outside the top-left rows x cols corner of n_planes contiguous (Hp, Wp) float32 planes := 0 (the border a windowed
 * call leaves untouched in a fresh gradient tensor). */
int uno_clear_border(float* t, long long n_planes, int Hp, int Wp, int rows, int cols, void* stream);
/* The end of the lift with the domain padding (reference darcy_flow_uno2d.py:100-107: `x_fc0 = self.fc0(x_fc); x_fc0 = F.gelu(x_fc0)`,
 * permute, `F.pad(x_fc0, [0, padding, 0, padding])`) in one pass: y_act (B, Co, Hp, Wp) = zero-pad(gelu(Wm [gelu](x (B, Ci, H, W)) + bias))
 * at the end of both axes, written by the layer's own store epilogue; y (B, Co, H, W) or NULL: the pre-activation result, kept only
 * if the caller wants it.  Backward without it: uno_channel_mix_dgelu_padded RECOMPUTES the layer (Ci << Co: reading x again is
 * half of what storing and re-reading y costs) and returns gz (B, Co, H, W) = gelu'(Wm [gelu](x) + bias) * g_padded[..., :H, :W] -
 * the gradient at the layer's output, ready for uno_channel_mix (transposed) / uno_channel_wgrad.
 * 260 <= W <= Wp, H * W < 2^24, float32. */
int uno_channel_mix_act_padded(const float* x, const float* w, const float* bias, float* y, float* y_act, int B, int Ci, int Co, int H,
                               int W, int Hp, int Wp, int act_in, void* stream);
int uno_channel_mix_dgelu_padded(const float* x, const float* w, const float* bias, const float* g_padded, float* gz, int B, int Ci,
                                 int Co, int H, int W, int Hp, int Wp, int act_in, void* stream);
/* The whole lift of the 2-D models (reference darcy_flow_uno2d.py:98-107): x (B, Cin <= 3, H, W) = [a(x, y), x, y] channels-first,
 *   h = fc_n1(x) (Cm = 16 or 32 channels; w1 (Cm, Cin), b1 (Cm) or NULL),  z = fc0(gelu(h)) (w0 (Co, Cm), b0 or NULL),
 *   act (B, Co, Hp, Wp) = zero-pad(gelu(z))
 * with NEITHER h nor z stored: h is 12 bytes per pixel of input against 128 of output, so every kernel that needs it evaluates it
 * from x.  With Cm = 32, Co = 64 the forward pass is one dedicated kernel that writes whole padded rows, and the backward pass is
 * one kernel per pixel tile (recomputed a = gelu(h) and z; gz = gelu'(z) g and gh = gelu'(h) w0^T gz stay on the chip, both layers'
 * weight-gradient sums are accumulated there) plus two small reductions; other widths run the generic kernels with gz and gh in
 * scratch.
 * uno_lift_backward: g_act (B, Co, Hp, Wp) -> gw1 (Cm, Cin), gb1 (Cm) or NULL, gw0 (Co, Cm), gb0 (Co) or NULL (written, not
 * accumulated; no gradient for x - it is data); ws: uno_lift_bwd_ws_bytes() bytes.  260 <= W <= Wp, H * W < 2^24, float32. */
int uno_lift_forward(const float* x, const float* w1, const float* b1, const float* w0, const float* b0, float* act, int B, int Cin,
                     int Cm, int Co, int H, int W, int Hp, int Wp, void* stream);
long long uno_lift_bwd_ws_bytes(int B, int Cin, int Cm, int Co, int H, int W);
int uno_lift_backward(const float* x, const float* w1, const float* b1, const float* w0, const float* b0, const float* g_act, float* gw1,
                      float* gb1, float* gw0, float* gb0, void* ws, int B, int Cin, int Cm, int Co, int H, int W, int Hp, int Wp,
                      void* stream);
/* ABI 11.  The same with a SECOND gradient of the padded activation, valid on the H x W domain of the same (Hp, Wp) planes: the lift's
 * output feeds two layers (conv0 and, through the skip connection, fc1: reference darcy_flow_uno2d.py:108-127), and the kernel that
 * streams the gradient adds the two as it reads them - no accumulation pass over the 64-channel tensor.  g_act2 may be NULL.
 * uno_lift_backward_takes_second: 1 where the one-kernel form runs (only it takes the second tensor). */
int uno_lift_backward_takes_second(int B, int Cin, int Cm, int Co, int H, int W, int Hp, int Wp);
int uno_lift_backward2(const float* x, const float* w1, const float* b1, const float* w0, const float* b0, const float* g_act,
                       const float* g_act2, float* gw1, float* gb1, float* gw0, float* gb0, void* ws, int B, int Cin, int Cm, int Co,
                       int H, int W, int Hp, int Wp, void* stream);

/* GELU followed by zero padding at the end of both axes (the lift's last activation + domain padding, reference
 * darcy_flow_uno2d.py:103-107): backward = 0: out (n_img, Hp, Wp) = pad(gelu(s (n_img, H, W))), gy ignored;
 * backward = 1: out (n_img, H, W) = gelu'(s) * gy[:, :H, :W] with gy (n_img, Hp, Wp). */
int uno_gelu_pad(const float* s, const float* gy, float* out, int n_img, int H, int W, int Hp, int Wp, int backward,
                 void* stream);

/* InstanceNorm (affine, biased variance, eps, no running statistics) optionally fused with the exact-erf GELU that follows
 * it in an operator block (reference integral_operators.py:269-270, 277-283; 3-D :497-498, 506-512).  x, y: (rows, N) with
 * rows = batch * C (channel of row r = r % C) and N = grid points; gamma, beta: (C) or NULL; mean, rstd: (rows) outputs kept
 * for the backward.  Backward: gx, and the per-row sums s1 = sum g_z, s2 = sum g_z * xhat (g_z = gradient at the affine
 * output) whose sums over the batch are the gradients of beta and gamma. */
int uno_instnorm_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long long rows,
                         int C, long long N, float eps, int gelu, void* stream);
int uno_instnorm_backward(const float* x, const float* gy, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, float* gx, float* s1, float* s2, long long rows, int C, long long N, int gelu,
                          void* stream);

/* One Adam update of one parameter tensor with the reference optimiser's semantics (Adam.py:27-52): coupled L2
 * weight decay (g += wd p) and, for complex tensors, the second moment from g conj(g) (one real entry per complex
 * entry).  p, g, m: float views (interleaved re/im when is_complex), v: n floats; n = entries (complex entries when
 * is_complex); step = 1-based step count for the bias corrections.  Updates p, m, v in place. */
int uno_adam_step(float* p, const float* g, float* m, float* v, long long n, int is_complex, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int step, void* stream);

/* The same update for a list of parameter tensors (host arrays of device pointers / sizes / flags): one call per
 * optimiser step instead of one per tensor. */
int uno_adam_step_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                        const long long* n, const int* is_complex, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int step, void* stream);

/* The same with the step count kept ON THE DEVICE (ABI 7), so that the update can be captured in a HIP graph and replayed:
 * `step_counter` (one int32, the number of steps taken so far: zero before the first step, the loaded count after a resume) is
 * advanced by one and the bias corrections lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) are evaluated in double by a one-thread kernel
 * into `scalars` (FOUR floats of device scratch owned by the caller: the two corrections, eps, weight_decay), which the update
 * kernel reads instead of kernel arguments.  Same arithmetic as uno_adam_step_multi with step = t.
 * ABI 9: `hyper` (NULL, or three doubles on the device: lr, eps, weight_decay) - when given, the three are read from the device at
 * execution time instead of from the arguments, so a learning-rate schedule (reference ns_train_2d.py:37,113) takes effect on
 * the next replay of a captured step: the caller rewrites the doubles between replays. */
int uno_adam_step_multi_dev(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                            const long long* n, const int* is_complex, double lr, double beta1, double beta2, double eps,
                            double weight_decay, int* step_counter, float* scalars, const double* hyper, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Mixed precision (BASELINE.json configs[4]: bf16 activations, half-precision weight storage, f32 accumulation).
 * The reference has no behaviour here (integral_operators.py:187 raises on bfloat16 input); the contract is "the float32
 * operator applied to the values the kernels read (inputs rounded once to bf16 / fp16), output rounded once to bf16".
 * `_bf16` entry points are their float32 namesakes with ACTIVATION tensors (x, y, grad tensors, `dgelu_of`) stored as
 * bfloat16 (void*); weights, biases, statistics, weight gradients, spectra and every accumulation stay float32 / complex64.
 * `_f16w` / `_mixed`: the complex weights are read as (re, im) float16 pairs; weight gradients come back complex64 (they
 * belong to the float32 master copy an optimiser keeps).
 */
int uno_spectral_conv2d_forward_mixed(const void* x_bf16, const void* w1_f16, const void* w2_f16, void* y_bf16,
                                      float* xtrunc, void* ws, int B, int Ci, int Co, int H, int W, int Ho, int Wo,
                                      int modes1, int modes2, void* stream);
int uno_spectral_conv2d_backward_mixed(const void* gy_bf16, const float* xtrunc, const void* w1_f16, const void* w2_f16,
                                       void* gx_bf16, float* gw1, float* gw2, void* ws, int B, int Ci, int Co, int H, int W,
                                       int Ho, int Wo, int modes1, int modes2, void* stream);
int uno_mode_mix_f16w(const float* in, const void* const* w_f16, float* out, int op, int B, int Ci, int Co, int ncorner,
                      int modes_per_corner, void* stream);
int uno_dft2d_forward_grouped_bf16(const void* images, float* spec, int n_img, int H, int W, int m1, int m2, float scale,
                                   int hermitian_cols, int mask_overlap, int group, int stride, int offset, void* stream);
int uno_dft2d_inverse_grouped_bf16(const float* spec, void* images, int n_img, int H, int W, int m1, int m2, float scale,
                                   int hermitian_cols, int mask_overlap, int group, int stride, int offset, void* stream);
int uno_resample2d_bf16(const void* in, void* out, float* tmp, int n_img, int H, int W, int Ho, int Wo, const int* startH,
                        const float* wtH, int KH, const int* startW, const float* wtW, int KW, const int* tile_p0,
                        const float* tile_w, int NP, int accumulate, void* stream);
int uno_channel_mix_bf16(const void* x, const float* w, const float* bias, void* y, int B, int Ci, int Co, long long P,
                         int transpose_w, int accumulate, int act_in, const void* dgelu_of, void* stream);
int uno_channel_wgrad_bf16(const void* gy, const void* x, float* gw, float* gb, void* ws, int B, int Ci, int Co, long long P,
                           int act_x, void* stream);
int uno_channel_mix2_bf16(const void* x1, const void* x2, int C1, const float* w, const float* bias, void* y1, void* y2, int Co1,
                          void* y_act, int B, int Ci, int Co, long long P, int transpose_w, int accumulate, int act_in,
                          const void* dgelu_of, const float* proj_w, const float* proj_b, void* proj_out, void* stream);
int uno_channel_wgrad2_bf16(const void* gy, const void* x1, const void* x2, int C1, float* gw, float* gb, void* ws, int B, int Ci,
                            int Co, long long P, int act_x, int accumulate, void* stream);
int uno_gelu_project_forward_bf16(const void* pre, const float* w, const float* bias, void* out, int B, int C, long long P,
                                  void* stream);
int uno_gelu_project_backward_bf16(const void* pre, const float* w, const void* gout, void* gpre, float* gw, float* gb,
                                   void* ws, int B, int C, long long P, void* stream);
int uno_gelu_pad_bf16(const void* s, const void* gy, void* out, int n_img, int H, int W, int Hp, int Wp, int backward,
                      void* stream);
int uno_instnorm_forward_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                              long long rows, int C, long long N, float eps, int gelu, void* stream);
int uno_instnorm_backward_bf16(const void* x, const void* gy, const float* gamma, const float* beta, const float* mean,
                               const float* rstd, void* gx, float* s1, float* s2, long long rows, int C, long long N,
                               int gelu, void* stream);

/* Scratch for the any-mode transforms (ABI 7).  Mode counts beyond the MFMA kernels' compiled range (modes1 > 40 or modes2 > 48 - the
 * reference's DEFAULT modes dim1//2 - 1, dim2//2, integral_operators.py:153-158) run two-pass plain-FMA transforms that need
 *   uno_dft2d_any_ws_bytes(n_img, H, W, m1, m2) = 8 n_img H m2 bytes of device scratch (0 inside the MFMA range)
 * per transform.  The library allocates nothing itself: the caller registers a device buffer for the calling THREAD with
 * uno_scratch_provide(ptr, bytes) before a call that may take that form (every uno_dft2d_* / uno_spectral_conv2d_* entry point; the
 * composite ones run their two transforms one after the other on the caller's stream, so max over the two sizes is enough), and clears it
 * with uno_scratch_provide(NULL, 0) afterwards.  The buffer must stay valid until the enqueued work has run (stream-ordered allocators:
 * free it on the same stream).  A call that needs more than was provided fails with -6 and a message naming the size. */
long long uno_dft2d_any_ws_bytes(int n_img, int H, int W, int m1, int m2);
int uno_scratch_provide(void* ptr, long long bytes);
/* ABI 10: the channel-mix entry points (uno_channel_mix*, uno_channel_mix2*) also look at the registered scratch.  Their wide layers
 * (>= 128 input channels; 32 on bfloat16 activations) run on three-piece bf16 operands; with uno_channel_mix_ws_bytes(Ci, Co, P, bf16)
 * = 6 Ci Co bytes of 16-byte aligned scratch (0: the shape never takes that form) the weights are split ONCE per call by a small
 * launch instead of by every workgroup.  Optional: without (enough) scratch the call runs as before, with identical results. */
long long uno_channel_mix_ws_bytes(int Ci, int Co, long long P, int bf16);

/* Batched transposing copy between the channels-last and the channels-first layout of an activation (ABI 8):
 *   out[b][c][r] = in[b][r][c],   b < B, r < R, c < C
 * in: R rows of C contiguous floats at pitch ld_in >= C, batch stride sb_in; out: C rows of R contiguous floats at pitch
 * ld_out >= R, batch stride sb_out (pitches and strides in floats).  The reference's blocks accept any strides because they go
 * through torch.fft / F.conv (integral_operators.py:187, 233); its model files hand the first block a channels-last tensor
 * (darcy_flow_uno2d.py:104-107: `permute` of the nn.Linear lift + F.pad; navier_stokes_uno3d.py:316-318) and their autograd
 * graph hands channels-last gradients back.  The kernels of this library read channels-first images, so the host side converts
 * with this entry point (R = pixels, C = channels: NHWC -> NCHW; R = channels, C = pixels: NCHW -> NHWC) instead of a generic
 * strided copy. */
int uno_transpose_batched(const float* in, float* out, int B, long long R, int C, long long ld_in, long long sb_in,
                          long long ld_out, long long sb_out, void* stream);

/* Optional in-library kernel timing (HIP events recorded on the launch stream around every kernel
 * this library enqueues), used by bench.py for the live roofline figure.
 *   uno_profile_begin(max_records): start recording (drops records beyond max_records).
 *   uno_profile_end():   stop, wait for the recorded events, return the number of records.
 *   uno_profile_get(i, name, name_len, &ms, &bytes): record i - kernel name as rocprofv3 prints it
 *     (e.g. "uno::dft2d_fwd_kernel<2, 3>"), its duration in ms and its ALGORITHMIC bytes
 *     (each operand counted once, DESIGN.md section "Roofline accounting"). */
int uno_profile_begin(int max_records);
int uno_profile_end(void);
int uno_profile_get(int index, char* name, int name_len, double* ms, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* UNO_SPECTRAL_H */
