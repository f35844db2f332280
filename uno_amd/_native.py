"""ctypes binding of libuno_spectral.so (C ABI: include/uno_spectral.h).

This is the host side of the drop-in boundary: plain pointers, sizes and a hipStream_t cross
it, nothing torch-typed.  There is NO fallback: if the library is missing or a call fails the
error is raised to the caller (RuntimeError, as the reference surfaces torch RuntimeErrors).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libuno_spectral.so")
ABI_VERSION = 12

_lib = None
_lock = threading.Lock()

_fp = C.c_void_p
_i = C.c_int
_SIGNATURES = {
    "uno_abi_version": (C.c_int, []),
    "uno_last_error": (C.c_char_p, []),
    "uno_dft2d_any_ws_bytes": (C.c_longlong, [_i] * 5),
    "uno_scratch_provide": (C.c_int, [_fp, C.c_longlong]),
    "uno_spectral_conv2d_fwd_ws_bytes": (C.c_longlong, [_i] * 5),
    "uno_spectral_conv2d_bwd_ws_bytes": (C.c_longlong, [_i] * 5),
    "uno_spectral_conv2d_forward": (C.c_int, [_fp] * 6 + [_i] * 9 + [_fp]),
    "uno_spectral_conv2d_backward": (C.c_int, [_fp] * 8 + [_i] * 9 + [_fp]),
    "uno_spectral_conv2d_forward_bf16": (C.c_int, [_fp] * 6 + [_i] * 9 + [_fp]),
    "uno_spectral_conv2d_backward_bf16": (C.c_int, [_fp] * 8 + [_i] * 9 + [_fp]),
    "uno_fft_resample3d_ws_bytes": (C.c_longlong, [_i] * 6),
    "uno_fft_resample3d": (C.c_int, [_fp] * 3 + [_i] * 8 + [_fp, _fp, _i, _fp, _fp, _i, C.c_float, _i, _i, _fp]),
    "uno_fft_resample3d_acc": (C.c_int, [_fp] * 4 + [_i] * 8 + [_fp, _fp, _i, _fp, _fp, _i, C.c_float, _i, _i, _fp]),
    "uno_dft2d_forward": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float, _i, _i, _fp]),
    "uno_dft2d_forward_bf16": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float, _i, _i, _fp]),
    "uno_dft2d_inverse_bf16": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float, _i, _i, _fp]),
    "uno_dft2d_inverse": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float, _i, _i, _fp]),
    "uno_reserve_cus": (C.c_int, [_i]),
    "uno_sweep_alternation": (C.c_int, [_i]),
    "uno_dft2d_inverse_add_applies": (C.c_int, [_i] * 7),
    "uno_dft2d_inverse_add": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float, _i, _i, _fp, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    "uno_dft2d_forward_grouped": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float] + [_i] * 5 + [_fp]),
    "uno_dft2d_inverse_grouped": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float] + [_i] * 5 + [_fp]),
    "uno_mode_mix": (C.c_int, [_fp, C.POINTER(_fp), _fp] + [_i] * 6 + [_fp]),
    "uno_mode_wgrad": (C.c_int, [_fp, _fp, C.POINTER(_fp)] + [_i] * 5 + [_fp]),
    "uno_mode_backward": (C.c_int, [_fp, _fp, C.POINTER(_fp), _fp, C.POINTER(_fp)] + [_i] * 6 + [_fp]),
    "uno_spectral_conv3d_fwd_ws_bytes": (C.c_longlong, [_i] * 8),
    "uno_spectral_conv3d_bwd_ws_bytes": (C.c_longlong, [_i] * 8),
    "uno_spectral_conv3d_forward": (C.c_int, [_fp, C.POINTER(_fp), _fp, _fp, _fp] + [_i] * 12 + [_fp]),
    "uno_spectral_conv3d_backward": (C.c_int, [_fp, _fp, C.POINTER(_fp), _fp, C.POINTER(_fp), _fp] + [_i] * 12 + [_fp]),
    "uno_cdft_axis": (C.c_int, [_fp, _fp] + [_i] * 6 + [C.c_float, _i, _fp]),
    "uno_resample2d": (C.c_int, [_fp, _fp, _fp] + [_i] * 5 + [_fp, _fp, _i, _fp, _fp, _i, _fp, _fp, _i, _i, _fp]),
    "uno_channel_mix": (C.c_int, [_fp, _fp, _fp, _fp, _i, _i, _i, C.c_longlong, _i, _i, _i, _fp, _fp]),
    "uno_channel_wgrad_ws_bytes": (C.c_longlong, [_i, _i, _i, C.c_longlong]),
    "uno_channel_wgrad": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, C.c_longlong, _i, _fp]),
    "uno_channel_mix2": (C.c_int, [_fp, _fp, _i, _fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, C.c_longlong, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    "uno_channel_wgrad2": (C.c_int, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _i, _i, _i, C.c_longlong, _i, _i, _fp]),
    "uno_channel_mix2_win": (C.c_int, [_fp, _fp, _i, _fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _i, _i, _i, C.c_longlong, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    "uno_channel_mix_ws_bytes": (C.c_longlong, [_i, _i, C.c_longlong, _i]),
    "uno_clear_border": (C.c_int, [_fp, C.c_longlong, _i, _i, _i, _i, _fp]),
    "uno_channel_mix_act_padded": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _fp]),
    "uno_lift_forward": (C.c_int, [_fp] * 6 + [_i] * 8 + [_fp]),
    "uno_lift_bwd_ws_bytes": (C.c_longlong, [_i] * 6),
    "uno_lift_backward": (C.c_int, [_fp] * 11 + [_i] * 8 + [_fp]),
    "uno_lift_backward2": (C.c_int, [_fp] * 12 + [_i] * 8 + [_fp]),
    "uno_lift_backward_takes_second": (C.c_int, [_i] * 8),
    "uno_channel_mix_dgelu_padded": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _fp]),
    "uno_channel_wgrad2_win": (C.c_int, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, C.c_longlong, _i, _i, _fp]),
    "uno_gelu_project_backward_win": (C.c_int, [_fp] * 7 + [_i, _i, _i, _i, _i, C.c_longlong, _fp]),
    "uno_channel_wgrad_finish": (C.c_int, [_fp, _fp, _fp, _i, _i, C.c_longlong, _i, _fp]),
    "uno_mode_wgrad_acc": (C.c_int, [_fp, _fp, C.POINTER(_fp)] + [_i] * 6 + [_fp]),
    "uno_spectral_conv2d_backward_acc": (C.c_int, [_fp] * 8 + [_i] * 11 + [_fp]),
    "uno_upload_table": (C.c_void_p, [_fp, C.c_longlong]),
    "uno_project_backward_applies": (C.c_int, [_i] * 7 + [C.c_longlong]),
    "uno_project_backward_ws_bytes": (C.c_longlong, [_i, _i, _i, C.c_longlong]),
    "uno_project_backward": (C.c_int, [_fp, _fp, _i] + [_fp] * 11 + [_i] * 6 + [C.c_longlong, _i, _i, _fp]),
    "uno_gelu_project_forward": (C.c_int, [_fp, _fp, _fp, _fp, _i, _i, C.c_longlong, _fp]),
    "uno_gelu_project_bwd_ws_bytes": (C.c_longlong, [_i, _i, C.c_longlong]),
    "uno_gelu_project_backward": (C.c_int, [_fp] * 7 + [_i, _i, C.c_longlong, _fp]),
    "uno_gelu_pad": (C.c_int, [_fp, _fp, _fp] + [_i] * 6 + [_fp]),
    "uno_transpose_batched": (C.c_int, [_fp, _fp, _i, C.c_longlong, _i] + [C.c_longlong] * 4 + [_fp]),
    "uno_instnorm_forward": (C.c_int, [_fp] * 6 + [C.c_longlong, _i, C.c_longlong, C.c_float, _i, _fp]),
    "uno_instnorm_backward": (C.c_int, [_fp] * 9 + [C.c_longlong, _i, C.c_longlong, _i, _fp]),
    "uno_adam_step": (C.c_int, [_fp, _fp, _fp, _fp, C.c_longlong, _i] + [C.c_double] * 5 + [_i, _fp]),
    "uno_adam_step_multi": (C.c_int, [_i, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(C.c_longlong),
                                      C.POINTER(_i)] + [C.c_double] * 5 + [_i, _fp]),
    "uno_adam_step_multi_dev": (C.c_int, [_i, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(C.c_longlong),
                                          C.POINTER(_i)] + [C.c_double] * 5 + [_fp, _fp, _fp, _fp]),
    "uno_spectral_conv2d_forward_mixed": (C.c_int, [_fp] * 6 + [_i] * 9 + [_fp]),
    "uno_spectral_conv2d_backward_mixed": (C.c_int, [_fp] * 8 + [_i] * 9 + [_fp]),
    "uno_mode_mix_f16w": (C.c_int, [_fp, C.POINTER(_fp), _fp] + [_i] * 6 + [_fp]),
    "uno_dft2d_forward_grouped_bf16": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float] + [_i] * 5 + [_fp]),
    "uno_dft2d_inverse_grouped_bf16": (C.c_int, [_fp, _fp] + [_i] * 5 + [C.c_float] + [_i] * 5 + [_fp]),
    "uno_resample2d_bf16": (C.c_int, [_fp, _fp, _fp] + [_i] * 5 + [_fp, _fp, _i, _fp, _fp, _i, _fp, _fp, _i, _i, _fp]),
    "uno_channel_mix_bf16": (C.c_int, [_fp, _fp, _fp, _fp, _i, _i, _i, C.c_longlong, _i, _i, _i, _fp, _fp]),
    "uno_channel_wgrad_bf16": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, C.c_longlong, _i, _fp]),
    "uno_channel_mix2_bf16": (C.c_int, [_fp, _fp, _i, _fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, C.c_longlong, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    "uno_channel_wgrad2_bf16": (C.c_int, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _i, _i, _i, C.c_longlong, _i, _i, _fp]),
    "uno_gelu_project_forward_bf16": (C.c_int, [_fp, _fp, _fp, _fp, _i, _i, C.c_longlong, _fp]),
    "uno_gelu_project_backward_bf16": (C.c_int, [_fp] * 7 + [_i, _i, C.c_longlong, _fp]),
    "uno_gelu_pad_bf16": (C.c_int, [_fp, _fp, _fp] + [_i] * 6 + [_fp]),
    "uno_instnorm_forward_bf16": (C.c_int, [_fp] * 6 + [C.c_longlong, _i, C.c_longlong, C.c_float, _i, _fp]),
    "uno_instnorm_backward_bf16": (C.c_int, [_fp] * 9 + [C.c_longlong, _i, C.c_longlong, _i, _fp]),
    "uno_profile_begin": (C.c_int, [_i]),
    "uno_profile_end": (C.c_int, []),
    "uno_profile_get": (C.c_int, [_i, C.c_char_p, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """Load (once) and return the native library; raises if it is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"uno_amd: native library {LIB_PATH} is missing - build it with "
                        "`python -m uno_amd.build` (hipcc, gfx950); there is no CPU fallback")
                h = C.CDLL(LIB_PATH)
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(h, name)
                    fn.restype = res
                    fn.argtypes = args
                if h.uno_abi_version() != ABI_VERSION:
                    raise RuntimeError(f"uno_amd: ABI mismatch, library {h.uno_abi_version()} != binding {ABI_VERSION}")
                _lib = h
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().uno_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def _ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _any_mode_scratch:
    """Scratch of the any-mode transforms (mode counts beyond the MFMA kernels' range: the reference's DEFAULT modes).  The library
    allocates nothing: `needs` = (n_img, H, W, m1, m2) of every transform the call may run; the largest requirement is taken from
    torch's caching allocator (stream-ordered: the block is reused only by later work on this stream), registered for this thread
    for the duration of the call and cleared afterwards."""

    def __init__(self, device, *needs):
        L = lib()
        self.bytes = max((int(L.uno_dft2d_any_ws_bytes(*[int(v) for v in n])) for n in needs), default=0)
        self.device = device
        self.buf = None

    def __enter__(self):
        if self.bytes > 0:
            self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=self.device)
            _check(lib().uno_scratch_provide(_ptr(self.buf), self.bytes), "uno_scratch_provide")
        return self

    def __exit__(self, *exc):
        if self.bytes > 0:
            lib().uno_scratch_provide(None, 0)
        return False


class _mix_scratch:
    """Scratch of the wide channel-mix layers (uno_channel_mix_ws_bytes: the weights pre-split for the bf16 matrix pipe), taken from
    torch's caching allocator and registered for this thread for the duration of the call - like _any_mode_scratch."""

    def __init__(self, device, Ci, Co, P, bf16):
        self.bytes = int(lib().uno_channel_mix_ws_bytes(int(Ci), int(Co), int(P), 1 if bf16 else 0))
        self.device = device
        self.buf = None

    def __enter__(self):
        if self.bytes > 0:
            self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=self.device)
            _check(lib().uno_scratch_provide(_ptr(self.buf), self.bytes), "uno_scratch_provide")
        return self

    def __exit__(self, *exc):
        if self.bytes > 0:
            lib().uno_scratch_provide(None, 0)
        return False


def _require(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"uno_amd: {name} must live on a HIP device (got {t.device}); the spectral "
                           "convolution runs only on the MI355X kernels")
    if t.dtype != dtype:
        raise RuntimeError(f"uno_amd: {name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"uno_amd: {name} must be contiguous")


def _require_dev(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"uno_amd: {name} must live on a HIP device (got {t.device})")
    if t.dtype != dtype:
        raise RuntimeError(f"uno_amd: {name} must be {dtype} (got {t.dtype})")


def _act_dtype(t, name):
    """f32, or bf16 for the mixed-precision entry points (activations bf16, everything else f32 / c64)."""
    bf16 = t.dtype == torch.bfloat16
    _require(t, torch.bfloat16 if bf16 else torch.float32, name)
    return bf16


def _weights_half(w1, w2, bf16):
    """True when the complex weights come as (..., 2) float16 (re, im) storage (mixed precision: bf16 activations only)."""
    if w1.dtype == torch.float16:
        if not bf16:
            raise RuntimeError("uno_amd: half-precision weight storage goes with bfloat16 activations")
        for w, name in ((w1, "weights1"), (w2, "weights2")):
            _require(w, torch.float16, name)
            if w.dim() != 5 or w.shape[-1] != 2:
                raise RuntimeError("uno_amd: half-precision weights are stored as (Ci, Co, m1, m2, 2) = (re, im)")
        return True
    _require(w1, torch.complex64, "weights1")
    _require(w2, torch.complex64, "weights2")
    return False


def spectral_conv2d_forward(x, w1, w2, Ho: int, Wo: int, xt_out=None):
    """-> (y (B,Co,Ho,Wo) in x's dtype (f32 | bf16), xtrunc (B,Ci,2*m1,m2) c64).  xt_out: dense complex64 tensor of that shape to
    write the truncated input spectrum into (a slot of a layer's time stack) instead of a fresh one."""
    bf16 = _act_dtype(x, "x")
    wh = _weights_half(w1, w2, bf16)
    B, Ci, H, W = x.shape
    Ci2, Co, m1, m2 = w1.shape[:4]
    if Ci2 != Ci or tuple(w2.shape) != tuple(w1.shape):
        raise RuntimeError(f"uno_amd: weight shapes {tuple(w1.shape)} / {tuple(w2.shape)} do not match input channels {Ci}")
    L = lib()
    with torch.cuda.device(x.device):
        y = torch.empty((B, Co, Ho, Wo), dtype=x.dtype, device=x.device)
        if xt_out is None:
            xt = torch.empty((B, Ci, 2 * m1, m2), dtype=torch.complex64, device=x.device)
        else:
            xt = xt_out
            _require(xt, torch.complex64, "xtrunc buffer")
            if tuple(xt.shape) != (B, Ci, 2 * m1, m2):
                raise RuntimeError("uno_amd: xtrunc buffer has the wrong shape")
        ws = torch.empty(max(1, L.uno_spectral_conv2d_fwd_ws_bytes(B, Ci, Co, m1, m2)), dtype=torch.uint8, device=x.device)
        fn = L.uno_spectral_conv2d_forward_mixed if wh else (L.uno_spectral_conv2d_forward_bf16 if bf16 else L.uno_spectral_conv2d_forward)
        with _any_mode_scratch(x.device, (B * Ci, H, W, m1, m2), (B * Co, Ho, Wo, m1, m2)):
            rc = fn(_ptr(x), _ptr(w1), _ptr(w2), _ptr(y), _ptr(xt), _ptr(ws), B, Ci, Co, H, W, Ho, Wo, m1, m2, _stream(x))
    _check(rc, "uno_spectral_conv2d_forward")
    return y, xt


def spectral_conv2d_backward(gy, xt, w1, w2, H: int, W: int, need_gx=True, need_gw=True, gw_out=None, accumulate_gw=False):
    """-> (gx or None (gy's dtype: f32 | bf16), gw1 or None, gw2 or None (c64)).
    gw_out = (gw1, gw2): write (accumulate_gw: add) the weight gradients into these complex64 tensors instead of fresh ones."""
    bf16 = _act_dtype(gy, "grad_output")
    _require(xt, torch.complex64, "xtrunc")
    wh = _weights_half(w1, w2, bf16)
    B, Co, Ho, Wo = gy.shape
    Ci, Co2, m1, m2 = w1.shape[:4]
    if Co2 != Co:
        raise RuntimeError("uno_amd: grad_output channels do not match the weights")
    L = lib()
    with torch.cuda.device(gy.device):
        gx = torch.empty((B, Ci, H, W), dtype=gy.dtype, device=gy.device) if need_gx else None
        if need_gw and gw_out is not None:
            gw1, gw2 = gw_out
            for t in gw_out:
                _require(t, torch.complex64, "weight-gradient buffer")
                if tuple(t.shape) != (Ci, Co, m1, m2):
                    raise RuntimeError("uno_amd: weight-gradient buffer has the wrong shape")
        else:
            accumulate_gw = False
            gw1 = torch.empty((Ci, Co, m1, m2), dtype=torch.complex64, device=gy.device) if need_gw else None
            gw2 = torch.empty((Ci, Co, m1, m2), dtype=torch.complex64, device=gy.device) if need_gw else None
        ws = torch.empty(max(1, L.uno_spectral_conv2d_bwd_ws_bytes(B, Ci, Co, m1, m2)), dtype=torch.uint8, device=gy.device)
        null = C.c_void_p(0)
        with _any_mode_scratch(gy.device, (B * Co, Ho, Wo, m1, m2), (B * Ci, H, W, m1, m2)):
            rc = L.uno_spectral_conv2d_backward_acc(_ptr(gy), _ptr(xt), _ptr(w1), _ptr(w2), _ptr(gx) if need_gx else null,
                                                    _ptr(gw1) if need_gw else null, _ptr(gw2) if need_gw else null,
                                                    _ptr(ws), B, Ci, Co, H, W, Ho, Wo, m1, m2, 2 if wh else (1 if bf16 else 0),
                                                    1 if accumulate_gw else 0, _stream(gy))
    _check(rc, "uno_spectral_conv2d_backward")
    return gx, gw1, gw2


def fft_resample3d(x, out_size, f1, f2, m3: int, scale: float, adjoint: bool, out=None, act=False):
    """x (..., D1, D2, D3) f32 -> (..., M1, M2, M3): pruned DFT with row frequencies f1[0] / f2[0] (int32 device tensors) along
    axes 1 / 2 and m3 half-spectrum bins, pruned inverse DFT with f1[1] / f2[1]; adjoint=True: Hermitian weights on the
    forward side instead of the inverse side (the transpose of the operator with sizes and tables swapped).
    out: ACCUMULATE into this (..., M1, M2, M3) tensor (returned); act: also return gelu(out) written in the same pass -> (out, act)."""
    _require(x, torch.float32, "x")
    *lead, D1, D2, D3 = x.shape
    M1, M2, M3 = (int(v) for v in out_size)
    n = 1
    for d in lead:
        n *= d
    for t in (*f1, *f2):
        if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("uno_amd: frequency tables must be contiguous int32 device tensors")
    J1, J2 = f1[0].numel(), f2[0].numel()
    L = lib()
    with torch.cuda.device(x.device):
        ws = torch.empty(max(1, L.uno_fft_resample3d_ws_bytes(n, D1, M1, J1, J2, int(m3))), dtype=torch.uint8, device=x.device)
        if out is not None:
            _require(out, torch.float32, "out")
            if tuple(out.shape) != (*lead, M1, M2, M3) or not out.is_contiguous():
                raise RuntimeError(f"uno_amd: out must be a contiguous {(*lead, M1, M2, M3)} tensor")
            ya = torch.empty_like(out) if act else None
            rc = L.uno_fft_resample3d_acc(_ptr(x), _ptr(out), _ptr(ya) if act else C.c_void_p(0), _ptr(ws), n, D1, D2, D3, M1, M2, M3,
                                          J1, _ptr(f1[0]), _ptr(f1[1]), J2, _ptr(f2[0]), _ptr(f2[1]), int(m3), float(scale),
                                          int(adjoint), int(not adjoint), _stream(x))
            _check(rc, "uno_fft_resample3d_acc")
            return (out, ya) if act else out
        y = torch.empty((*lead, M1, M2, M3), dtype=torch.float32, device=x.device)
        rc = L.uno_fft_resample3d(_ptr(x), _ptr(y), _ptr(ws), n, D1, D2, D3, M1, M2, M3, J1, _ptr(f1[0]), _ptr(f1[1]), J2,
                                  _ptr(f2[0]), _ptr(f2[1]), int(m3), float(scale), int(adjoint), int(not adjoint), _stream(x))
    _check(rc, "uno_fft_resample3d")
    return y


def dft2d_forward(images, m1, m2, scale=1.0, hermitian_cols=False, mask_overlap=False, out=None, channel_offset=0):
    """images (..., H, W) f32 -> spectra (..., 2*m1, m2) c64.  With `out` (B, Ctot, 2*m1, m2) and images (B, C1, H, W) the
    spectra go to channels [channel_offset, channel_offset + C1) of `out`.  bf16 images: plain form only."""
    bf16 = _act_dtype(images, "images")
    *lead, H, W = images.shape
    n = 1
    for d in lead:
        n *= d
    if out is None:
        spec = torch.empty((*lead, 2 * m1, m2), dtype=torch.complex64, device=images.device)
        with torch.cuda.device(images.device):
            fn = lib().uno_dft2d_forward_bf16 if bf16 else lib().uno_dft2d_forward
            with _any_mode_scratch(images.device, (n, H, W, m1, m2)):
                rc = fn(_ptr(images), _ptr(spec), n, H, W, m1, m2, float(scale), int(hermitian_cols), int(mask_overlap), _stream(images))
        _check(rc, "uno_dft2d_forward")
        return spec
    _require(out, torch.complex64, "out")
    if images.dim() != 4 or out.dim() != 4 or out.shape[0] != images.shape[0] or tuple(out.shape[2:]) != (2 * m1, m2) \
            or channel_offset < 0 or channel_offset + images.shape[1] > out.shape[1]:
        raise RuntimeError("uno_amd: out must be (B, Ctot, 2*m1, m2) with room for the image channels at channel_offset")
    with torch.cuda.device(images.device):
        fn = lib().uno_dft2d_forward_grouped_bf16 if bf16 else lib().uno_dft2d_forward_grouped
        with _any_mode_scratch(images.device, (n, H, W, m1, m2)):
            rc = fn(_ptr(images), _ptr(out), n, H, W, m1, m2, float(scale), int(hermitian_cols),
                    int(mask_overlap), images.shape[1], out.shape[1], int(channel_offset), _stream(images))
    _check(rc, "uno_dft2d_forward_grouped")
    return out


def reserve_cus(n: int) -> int:
    """Set aside n compute units for communication kernels running beside the library's (uno_reserve_cus); returns the previous value."""
    return int(lib().uno_reserve_cus(int(n)))


def sweep_alternation(enable: bool) -> bool:
    """Alternating sweep direction of consecutive launches (uno_sweep_alternation); returns the previous setting."""
    return bool(lib().uno_sweep_alternation(1 if enable else 0))


def dft2d_inverse_add_applies(n_img, H, W, m1, m2, Hs, Ws) -> bool:
    """True where dft2d_inverse(..., addend=) runs the fused kernel (K3 + up-sampled addend, csrc/dft2d_inv_add_kernel.h)."""
    return bool(lib().uno_dft2d_inverse_add_applies(int(n_img), int(H), int(W), int(m1), int(m2), int(Hs), int(Ws)))


def dft2d_inverse(spec, H, W, scale=1.0, hermitian_cols=True, mask_overlap=True, channels=None, channel_offset=0,
                  dtype=torch.float32, addend=None):
    """spectra (..., 2*m1, m2) c64 -> images (..., H, W) f32 (or bf16 with dtype=torch.bfloat16, plain form only).  With `channels` = C1 and spec (B, Ctot, 2*m1, m2) only the
    channels [channel_offset, channel_offset + C1) are transformed -> (B, C1, H, W).
    addend = (t (..., Hs, Ws) f32, (tile_p0, row_op, col_v0, col_op)): images += the banded up-sampling of t, in the same pass
    (uno_dft2d_inverse_add; the caller checked dft2d_inverse_add_applies)."""
    _require(spec, torch.complex64, "spec")
    *lead, r2, m2 = spec.shape
    m1 = r2 // 2
    if addend is not None:
        t, (p0, rowop, v0, colop) = addend
        _require(t, torch.float32, "addend")
        if channels is not None or dtype != torch.float32 or tuple(t.shape[:-2]) != tuple(lead):
            raise RuntimeError("uno_amd: dft2d_inverse(addend=) takes the plain float32 form with one addend image per spectrum")
        n = 1
        for d in lead:
            n *= d
        img = torch.empty((*lead, H, W), dtype=torch.float32, device=spec.device)
        with torch.cuda.device(spec.device):
            rc = lib().uno_dft2d_inverse_add(_ptr(spec), _ptr(img), n, H, W, m1, m2, float(scale), int(hermitian_cols), int(mask_overlap),
                                             _ptr(t), t.shape[-2], t.shape[-1], _ptr(p0), _ptr(rowop), _ptr(v0), _ptr(colop), _stream(spec))
        _check(rc, "uno_dft2d_inverse_add")
        return img
    if channels is None:
        n = 1
        for d in lead:
            n *= d
        if dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError("uno_amd: images are float32 or bfloat16")
        img = torch.empty((*lead, H, W), dtype=dtype, device=spec.device)
        with torch.cuda.device(spec.device):
            fn = lib().uno_dft2d_inverse_bf16 if dtype == torch.bfloat16 else lib().uno_dft2d_inverse
            with _any_mode_scratch(spec.device, (n, H, W, m1, m2)):
                rc = fn(_ptr(spec), _ptr(img), n, H, W, m1, m2, float(scale), int(hermitian_cols), int(mask_overlap), _stream(spec))
        _check(rc, "uno_dft2d_inverse")
        return img
    if dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("uno_amd: images are float32 or bfloat16")
    if spec.dim() != 4 or channel_offset < 0 or channels < 1 or channel_offset + channels > spec.shape[1]:
        raise RuntimeError("uno_amd: spec must be (B, Ctot, 2*m1, m2) holding the requested channel range")
    B = spec.shape[0]
    img = torch.empty((B, channels, H, W), dtype=dtype, device=spec.device)
    with torch.cuda.device(spec.device):
        fn = lib().uno_dft2d_inverse_grouped_bf16 if dtype == torch.bfloat16 else lib().uno_dft2d_inverse_grouped
        with _any_mode_scratch(spec.device, (B * channels, H, W, m1, m2)):
            rc = fn(_ptr(spec), _ptr(img), B * channels, H, W, m1, m2, float(scale), int(hermitian_cols),
                    int(mask_overlap), int(channels), spec.shape[1], int(channel_offset), _stream(spec))
    _check(rc, "uno_dft2d_inverse_grouped")
    return img


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def mode_mix(inp, weights, op: int):
    """inp (B, Cin, ncorner, modes) c64; weights: list of (Ci, Co, modes...) c64."""
    _require(inp, torch.complex64, "spectrum")
    half = weights[0].dtype == torch.float16
    for w in weights:
        _require(w, torch.float16 if half else torch.complex64, "weights")
    B = inp.shape[0]
    Ci, Co = weights[0].shape[:2]
    nc = len(weights)
    Mc = weights[0][0, 0].numel() // (2 if half else 1)
    cout = Co if op == 0 else Ci
    out = torch.empty((B, cout, *inp.shape[2:]), dtype=torch.complex64, device=inp.device)
    with torch.cuda.device(inp.device):
        fn = lib().uno_mode_mix_f16w if half else lib().uno_mode_mix
        rc = fn(_ptr(inp), _ptr_array(weights), _ptr(out), op, B, Ci, Co, nc, Mc, _stream(inp))
    _check(rc, "uno_mode_mix")
    return out


def mode_wgrad(xt, go, weight_shape, ncorner: int, out=None, accumulate: bool = False):
    """out: list of `ncorner` complex64 tensors of `weight_shape` to write (accumulate: add) the gradients into."""
    _require(xt, torch.complex64, "xtrunc")
    _require(go, torch.complex64, "grad spectrum")
    B, Ci = xt.shape[:2]
    Co = go.shape[1]
    if out is None:
        accumulate = False
        gws = [torch.empty(weight_shape, dtype=torch.complex64, device=xt.device) for _ in range(ncorner)]
    else:
        gws = list(out)
        for t in gws:
            _require(t, torch.complex64, "weight-gradient buffer")
            if tuple(t.shape) != tuple(weight_shape):
                raise RuntimeError("uno_amd: weight-gradient buffer has the wrong shape")
    Mc = gws[0][0, 0].numel()
    with torch.cuda.device(xt.device):
        rc = lib().uno_mode_wgrad_acc(_ptr(xt), _ptr(go), _ptr_array(gws), B, Ci, Co, ncorner, Mc, 1 if accumulate else 0, _stream(xt))
    _check(rc, "uno_mode_wgrad")
    return gws


def mode_backward(xt, go, weights, out=None, accumulate: bool = False):
    """Both per-mode GEMMs of a backward pass in one launch (uno_mode_backward): xt (B, Ci, ncorner, modes) and go (B, Co, ncorner, modes)
    c64 (any trailing shape with ncorner * modes entries per channel), weights: list of ncorner (Ci, Co, modes...) c64
    -> (gX (B, Ci, ncorner * modes) c64, [gw per corner]); out / accumulate as in mode_wgrad."""
    _require(xt, torch.complex64, "xtrunc")
    _require(go, torch.complex64, "grad spectrum")
    for w in weights:
        _require(w, torch.complex64, "weights")
    B, Ci = xt.shape[:2]
    Co = go.shape[1]
    nc = len(weights)
    Mc = weights[0][0, 0].numel()
    if out is None:
        accumulate = False
        gws = [torch.empty(weights[0].shape, dtype=torch.complex64, device=xt.device) for _ in range(nc)]
    else:
        gws = list(out)
        for t in gws:
            _require(t, torch.complex64, "weight-gradient buffer")
            if tuple(t.shape) != tuple(weights[0].shape):
                raise RuntimeError("uno_amd: weight-gradient buffer has the wrong shape")
    gX = torch.empty((B, Ci, nc * Mc), dtype=torch.complex64, device=xt.device)
    with torch.cuda.device(xt.device):
        rc = lib().uno_mode_backward(_ptr(xt), _ptr(go), _ptr_array(list(weights)), _ptr(gX), _ptr_array(gws), B, Ci, Co, nc, Mc,
                                     1 if accumulate else 0, _stream(xt))
    _check(rc, "uno_mode_backward")
    return gX, gws


def spectral_conv3d_forward(x, ws_, Ho: int, Wo: int, To: int):
    """x (B,Ci,H,W,T) f32, ws_ = [weights1..4] (Ci,Co,m1,m2,m3) c64 -> (y, xtrunc (B,Ci,4,m1,m2,m3) c64)."""
    _require(x, torch.float32, "x")
    if len(ws_) != 4:
        raise RuntimeError("uno_amd: the 3-D spectral convolution takes four weight tensors")
    for w in ws_:
        _require(w, torch.complex64, "weights")
        if tuple(w.shape) != tuple(ws_[0].shape):
            raise RuntimeError("uno_amd: weights1..4 must have one shape")
    B, Ci, H, W, T = x.shape
    Ci2, Co, m1, m2, m3 = ws_[0].shape
    if Ci2 != Ci:
        raise RuntimeError(f"uno_amd: weight shape {tuple(ws_[0].shape)} does not match input channels {Ci}")
    L = lib()
    with torch.cuda.device(x.device):
        y = torch.empty((B, Co, Ho, Wo, To), dtype=torch.float32, device=x.device)
        xt = torch.empty((B, Ci, 4, m1, m2, m3), dtype=torch.complex64, device=x.device)
        scratch = torch.empty(max(1, L.uno_spectral_conv3d_fwd_ws_bytes(B, Ci, Co, H, Ho, m1, m2, m3)), dtype=torch.uint8,
                              device=x.device)
        # (the (W, T) planes go through the 2-D transforms with modes (m2, m3): beyond the MFMA range they take the any-mode form)
        with _any_mode_scratch(x.device, (B * Ci * H, W, T, m2, m3), (B * Co * Ho, Wo, To, m2, m3)):
            rc = L.uno_spectral_conv3d_forward(_ptr(x), _ptr_array(ws_), _ptr(y), _ptr(xt), _ptr(scratch), B, Ci, Co,
                                               H, W, T, Ho, Wo, To, m1, m2, m3, _stream(x))
    _check(rc, "uno_spectral_conv3d_forward")
    return y, xt


def spectral_conv3d_backward(gy, xt, ws_, H: int, W: int, T: int, need_gx=True, need_gw=True):
    _require(gy, torch.float32, "grad_output")
    _require(xt, torch.complex64, "xtrunc")
    for w in ws_:
        _require(w, torch.complex64, "weights")
    B, Co, Ho, Wo, To = gy.shape
    Ci, Co2, m1, m2, m3 = ws_[0].shape
    if Co2 != Co:
        raise RuntimeError("uno_amd: grad_output channels do not match the weights")
    L = lib()
    with torch.cuda.device(gy.device):
        gx = torch.empty((B, Ci, H, W, T), dtype=torch.float32, device=gy.device) if need_gx else None
        gws = [torch.empty_like(w) for w in ws_] if need_gw else None
        scratch = torch.empty(max(1, L.uno_spectral_conv3d_bwd_ws_bytes(B, Ci, Co, H, Ho, m1, m2, m3)), dtype=torch.uint8,
                              device=gy.device)
        with _any_mode_scratch(gy.device, (B * Co * Ho, Wo, To, m2, m3), (B * Ci * H, W, T, m2, m3)):
            rc = L.uno_spectral_conv3d_backward(_ptr(gy), _ptr(xt), _ptr_array(ws_), _ptr(gx) if need_gx else C.c_void_p(0),
                                                _ptr_array(gws) if need_gw else None, _ptr(scratch), B, Ci, Co,
                                                H, W, T, Ho, Wo, To, m1, m2, m3, _stream(gy))
    _check(rc, "uno_spectral_conv3d_backward")
    return gx, gws


def resample2d(x, Ho: int, Wo: int, tabH, tabW, tilesH=None, out=None, reverse: bool = False):
    """x (..., H, W) f32 -> (..., Ho, Wo); tabX = (start int32 [out], weights f32 [out, K]) band tables on x.device;
    tilesH = (p0 int32 [ntiles], dense weights f32 [ntiles, NP, 16]) enables the fused single-pass kernel.
    out: accumulate into this (..., Ho, Wo) tensor instead of allocating the result.
    reverse: walk the images in descending order (same result; see csrc/resample2d.hip)."""
    bf16 = _act_dtype(x, "x")
    *lead, H, W = x.shape
    n = 1
    for d in lead:
        n *= d
    sH, wH = tabH
    sW, wW = tabW
    accumulate = out is not None
    if accumulate:
        _require(out, x.dtype, "out")
        if tuple(out.shape) != (*lead, Ho, Wo):
            raise RuntimeError(f"uno_amd: out has shape {tuple(out.shape)}, expected {(*lead, Ho, Wo)}")
    else:
        out = torch.empty((*lead, Ho, Wo), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        tmp = torch.empty(max(1, n * min(Ho * W, H * Wo)), dtype=torch.float32, device=x.device)
        if tilesH is not None:
            tp0, tw = tilesH
            targs = (_ptr(tp0), _ptr(tw), tw.shape[1])
        else:
            targs = (C.c_void_p(0), C.c_void_p(0), 0)
        fn = lib().uno_resample2d_bf16 if bf16 else lib().uno_resample2d
        rc = fn(_ptr(x), _ptr(out), _ptr(tmp), n, H, W, Ho, Wo, _ptr(sH), _ptr(wH), wH.shape[1],
                _ptr(sW), _ptr(wW), wW.shape[1], *targs, (1 if accumulate else 0) | (2 if reverse else 0), _stream(x))
    _check(rc, "uno_resample2d")
    return out


def _window_args(window, plane: int, bf16: bool):
    """(rows, cols, pitch) of a windowed call (uno_spectral.h, ABI 10): the tensors' last axis is a whole plane of `plane` elements."""
    rows, cols, pitch = (int(v) for v in window)
    if bf16:
        raise RuntimeError("uno_amd: pixel windows are float32 only")
    if rows < 1 or cols % 4 or cols < 260 or pitch < cols or (rows - 1) * pitch + cols > plane or rows * cols >= 1 << 24:
        raise RuntimeError(f"uno_amd: window rows={rows} cols={cols} pitch={pitch} does not fit a plane of {plane} elements "
                           "(cols: a multiple of 4, 260 <= cols <= pitch; rows * cols < 2^24)")
    return rows, cols, pitch


def channel_mix(x, w, bias=None, transpose_w: bool = False, out=None, act_in: bool = False, dgelu_of=None, dgelu_total: bool = False,
                window=None):
    """x (B, Ci, P) f32, w (Co, Ci) (or (Ci, Co) with transpose_w) -> y (B, Co, P) = Wm x + bias;
    out: accumulate into this (B, Co, P) tensor instead; act_in: x := gelu(x) as it is read; dgelu_of (B, Co, P): the
    product is multiplied by gelu'(dgelu_of) - with dgelu_total (and out) the completed sum out + product is.
    window = (rows, cols, pitch): the call works on that window of planes of P elements (see channel_mix2)."""
    if window is not None:
        return channel_mix2(x, None, w, bias, transpose_w=transpose_w, out=out, act_in=act_in, dgelu_of=dgelu_of,
                            accumulate=(2 if (dgelu_total and dgelu_of is not None) else 1) if out is not None else 0, window=window)
    bf16 = _act_dtype(x, "x")
    _require(w, torch.float32, "weight")
    if bias is not None:
        _require(bias, torch.float32, "bias")
    B, Ci, P = x.shape
    Co = w.shape[1] if transpose_w else w.shape[0]
    if (w.shape[0] if transpose_w else w.shape[1]) != Ci:
        raise RuntimeError(f"uno_amd: weight {tuple(w.shape)} does not match {Ci} input channels")
    accumulate = out is not None
    if accumulate:
        _require(out, x.dtype, "out")
        if tuple(out.shape) != (B, Co, P):
            raise RuntimeError(f"uno_amd: out has shape {tuple(out.shape)}, expected {(B, Co, P)}")
        y = out
    else:
        y = torch.empty((B, Co, P), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _mix_scratch(x.device, Ci, Co, P, bf16):
        if dgelu_of is not None:
            _require(dgelu_of, x.dtype, "dgelu_of")
            if tuple(dgelu_of.shape) != (B, Co, P):
                raise RuntimeError(f"uno_amd: dgelu_of has shape {tuple(dgelu_of.shape)}, expected {(B, Co, P)}")
        fn = lib().uno_channel_mix_bf16 if bf16 else lib().uno_channel_mix
        rc = fn(_ptr(x), _ptr(w), _ptr(bias) if bias is not None else C.c_void_p(0), _ptr(y),
                                   B, Ci, Co, P, 1 if transpose_w else 0,
                                   (2 if (dgelu_total and dgelu_of is not None) else 1) if accumulate else 0, 1 if act_in else 0,
                                   _ptr(dgelu_of) if dgelu_of is not None else C.c_void_p(0), _stream(x))
    _check(rc, "uno_channel_mix")
    return y


def clear_border(t, rows: int, cols: int):
    """t (..., Hp, Wp) float32 contiguous: everything outside t[..., :rows, :cols] := 0, in place (-> t)."""
    _require(t, torch.float32, "tensor")
    Hp, Wp = t.shape[-2:]
    n = t.numel() // max(1, Hp * Wp)
    with torch.cuda.device(t.device):
        rc = lib().uno_clear_border(_ptr(t), n, Hp, Wp, int(rows), int(cols), _stream(t))
    _check(rc, "uno_clear_border")
    return t


def channel_mix_act_padded_ok(x, Hp: int, Wp: int) -> bool:
    """shape rules of uno_channel_mix_act_padded for x (B, Ci, H, W)"""
    H, W = x.shape[-2:]
    return x.dtype == torch.float32 and 260 <= W <= Wp and H <= Hp and H * W < (1 << 24)


def channel_mix_act_padded(x, w, bias, Hp: int, Wp: int, act_in: bool = False, keep_y: bool = True):
    """x (B, Ci, H, W) f32, w (Co, Ci) -> y (B, Co, H, W) = w . [gelu](x) + bias (None with keep_y=False: not stored) and
    act (B, Co, Hp, Wp) = zero-pad(gelu(y))."""
    _require(x, torch.float32, "x")
    _require(w, torch.float32, "weight")
    if bias is not None:
        _require(bias, torch.float32, "bias")
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    if w.shape[1] != Ci:
        raise RuntimeError(f"uno_amd: weight {tuple(w.shape)} does not match {Ci} input channels")
    y = torch.empty((B, Co, H, W), dtype=x.dtype, device=x.device) if keep_y else None
    act = torch.empty((B, Co, Hp, Wp), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib().uno_channel_mix_act_padded(_ptr(x), _ptr(w), _ptr(bias) if bias is not None else C.c_void_p(0),
                                              _ptr(y) if keep_y else C.c_void_p(0), _ptr(act),
                                              B, Ci, Co, H, W, int(Hp), int(Wp), 1 if act_in else 0, _stream(x))
    _check(rc, "uno_channel_mix_act_padded")
    return y, act


def lift_ok(x, w1, w0, Hp: int, Wp: int) -> bool:
    """shape rules of uno_lift_forward / uno_lift_backward for x (B, Cin, H, W)"""
    if x.dim() != 4 or x.dtype != torch.float32:
        return False
    H, W = x.shape[-2:]
    return (1 <= x.shape[1] <= 3 and w1.shape[0] in (16, 32) and w1.shape[1] == x.shape[1] and w0.shape[1] == w1.shape[0]
            and 260 <= W <= Wp and H <= Hp and H * W < (1 << 24))


def lift_forward(x, w1, b1, w0, b0, Hp: int, Wp: int):
    """act (B, Co, Hp, Wp) = zero-pad(gelu(w0 . gelu(w1 . x + b1) + b0)) - nothing else is stored (uno_lift_forward)."""
    for t, n in ((x, "x"), (w1, "weight 1"), (w0, "weight 0")) + tuple((t, "bias") for t in (b1, b0) if t is not None):
        _require(t, torch.float32, n)
    B, Cin, H, W = x.shape
    Cm, Co = w1.shape[0], w0.shape[0]
    act = torch.empty((B, Co, Hp, Wp), dtype=x.dtype, device=x.device)
    null = C.c_void_p(0)
    with torch.cuda.device(x.device):
        rc = lib().uno_lift_forward(_ptr(x), _ptr(w1), _ptr(b1) if b1 is not None else null, _ptr(w0), _ptr(b0) if b0 is not None else null,
                                    _ptr(act), B, Cin, Cm, Co, H, W, int(Hp), int(Wp), _stream(x))
    _check(rc, "uno_lift_forward")
    return act


def lift_backward_takes_second(x, w1, w0, Hp: int, Wp: int) -> bool:
    B, Cin, H, W = x.shape
    return bool(lib().uno_lift_backward_takes_second(B, Cin, w1.shape[0], w0.shape[0], H, W, int(Hp), int(Wp)))


def lift_backward(x, w1, b1, w0, b0, g_act, g_act2=None):
    """-> gw1 (Cm, Cin), gb1 (Cm) or None, gw0 (Co, Cm), gb0 (Co) or None (uno_lift_backward / uno_lift_backward2).
    g_act2: a second gradient of the padded activation (same shape, valid on the domain) added as the kernel reads (fused kernel only)."""
    _require(g_act, torch.float32, "grad_output")
    B, Cin, H, W = x.shape
    Cm, Co = w1.shape[0], w0.shape[0]
    Hp, Wp = g_act.shape[-2:]
    if tuple(g_act.shape[:2]) != (B, Co):
        raise RuntimeError("uno_amd: grad_output does not match the lift")
    if g_act2 is not None:
        _require(g_act2, torch.float32, "second grad_output")
        if tuple(g_act2.shape) != tuple(g_act.shape):
            raise RuntimeError("uno_amd: the two gradients of the lift's output disagree in shape")
    dev = x.device
    gw1 = torch.empty((Cm, Cin), dtype=torch.float32, device=dev)
    gw0 = torch.empty((Co, Cm), dtype=torch.float32, device=dev)
    gb1 = torch.empty((Cm,), dtype=torch.float32, device=dev) if b1 is not None else None
    gb0 = torch.empty((Co,), dtype=torch.float32, device=dev) if b0 is not None else None
    null = C.c_void_p(0)
    with torch.cuda.device(dev):
        ws = torch.empty(max(1, lib().uno_lift_bwd_ws_bytes(B, Cin, Cm, Co, H, W)), dtype=torch.uint8, device=dev)
        rc = lib().uno_lift_backward2(_ptr(x), _ptr(w1), _ptr(b1) if b1 is not None else null, _ptr(w0), _ptr(b0) if b0 is not None else null,
                                     _ptr(g_act), _ptr(g_act2) if g_act2 is not None else null, _ptr(gw1), _ptr(gb1) if gb1 is not None else null, _ptr(gw0),
                                     _ptr(gb0) if gb0 is not None else null, _ptr(ws), B, Cin, Cm, Co, H, W, int(Hp), int(Wp), _stream(x))
    _check(rc, "uno_lift_backward")
    return gw1, gb1, gw0, gb0


def channel_mix_dgelu_padded(x, w, bias, g_padded, act_in: bool = False):
    """gz (B, Co, H, W) = gelu'(w . [gelu](x) + bias) * g_padded[..., :H, :W]: the layer recomputed from its input x (B, Ci, H, W)."""
    _require(x, torch.float32, "x")
    _require(w, torch.float32, "weight")
    _require(g_padded, torch.float32, "grad_output")
    if bias is not None:
        _require(bias, torch.float32, "bias")
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    if w.shape[1] != Ci or g_padded.dim() != 4 or g_padded.shape[:2] != (B, Co) or g_padded.shape[2] < H or g_padded.shape[3] < W:
        raise RuntimeError("uno_amd: weight / grad_output do not match the layer")
    Hp, Wp = g_padded.shape[2:]
    gz = torch.empty((B, Co, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib().uno_channel_mix_dgelu_padded(_ptr(x), _ptr(w), _ptr(bias) if bias is not None else C.c_void_p(0), _ptr(g_padded), _ptr(gz),
                                                B, Ci, Co, H, W, int(Hp), int(Wp), 1 if act_in else 0, _stream(x))
    _check(rc, "uno_channel_mix_dgelu_padded")
    return gz


def channel_mix2_ok(C1: int, Co1, Co: int, P: int) -> bool:
    """shape rules of the fused two-source / two-destination forms (csrc/channel_mix.hip)"""
    if C1 is not None and (C1 < 16 or C1 % 16):
        return False
    if Co1 is not None and (Co1 < 64 or Co1 % 64):
        return False
    return True


def channel_mix2(x1, x2, w, bias=None, transpose_w: bool = False, out=None, out2=None, split_out: int | None = None,
                 act_in: bool = False, dgelu_of=None, y_act: bool = False, accumulate: bool = False, project=None, window=None):
    """uno_channel_mix2: y = Wm . cat(x1, x2) + bias in one pass (x2 may be None).
    split_out = Co1: the output channels go to two tensors (B, Co1, P), (B, Co - Co1, P) -> returns (y1, y2);
    y_act: also return gelu(y) as a second tensor -> (y, act); act_in / dgelu_of act on x1 / y1 only;
    out (and out2): write (accumulate=True: add) into the given tensors instead of allocating;
    project = (w2 (Co,), b2 (1,) or None): also return proj (B, P) = b2 + sum_o w2[o] gelu(y[:, o]) -> (y, proj)  (Co <= 64).
    window = (rows, cols, pitch): every tensor's last axis is a whole plane of P elements of which the call reads and writes only
    the window - rows x cols points, row r at r * pitch (uno_channel_mix2_win; float32; fresh outputs are torch.empty planes whose
    elements outside the window are left as they are).  accumulate = 2 (with out and dgelu_of, one destination): gelu' multiplies
    the completed sum."""
    bf16 = _act_dtype(x1, "x1")
    _require(w, torch.float32, "weight")
    if x2 is not None:
        _require(x2, x1.dtype, "x2")
        if x2.shape[0] != x1.shape[0] or x2.shape[2] != x1.shape[2]:
            raise RuntimeError("uno_amd: the two sources disagree in batch / pixel count")
    if bias is not None:
        _require(bias, torch.float32, "bias")
    B, C1, P = x1.shape
    Ci = C1 + (x2.shape[1] if x2 is not None else 0)
    Co = w.shape[1] if transpose_w else w.shape[0]
    if (w.shape[0] if transpose_w else w.shape[1]) != Ci:
        raise RuntimeError(f"uno_amd: weight {tuple(w.shape)} does not match {Ci} input channels")
    Co1 = Co if split_out is None else int(split_out)
    if out is None:
        y1 = torch.empty((B, Co1, P), dtype=x1.dtype, device=x1.device)
        y2 = torch.empty((B, Co - Co1, P), dtype=x1.dtype, device=x1.device) if split_out is not None else None
    else:
        y1, y2 = out, out2
        _require(y1, x1.dtype, "out")
        if tuple(y1.shape) != (B, Co1, P) or (split_out is not None and (y2 is None or tuple(y2.shape) != (B, Co - Co1, P))):
            raise RuntimeError("uno_amd: output tensors do not match (B, Co1, P) / (B, Co - Co1, P)")
        if y2 is not None:
            _require(y2, x1.dtype, "out2")
    act = torch.empty((B, Co, P), dtype=x1.dtype, device=x1.device) if y_act else None
    if dgelu_of is not None:
        _require(dgelu_of, x1.dtype, "dgelu_of")
        if tuple(dgelu_of.shape) != (B, Co1, P):
            raise RuntimeError(f"uno_amd: dgelu_of has shape {tuple(dgelu_of.shape)}, expected {(B, Co1, P)}")
    null = C.c_void_p(0)
    proj = pw = pb = None
    if project is not None:
        pw, pb = project
        _require(pw, torch.float32, "projection weight")
        if pw.numel() != Co:
            raise RuntimeError(f"uno_amd: projection weight has {pw.numel()} entries for {Co} channels")
        if pb is not None:
            _require(pb, torch.float32, "projection bias")
        proj = torch.empty((B, P), dtype=x1.dtype, device=x1.device)
    acc_flag = int(accumulate) if out is not None else 0
    if acc_flag == 2 and (dgelu_of is None or window is None):
        raise RuntimeError("uno_amd: accumulate = 2 goes with dgelu_of (windowed one-destination calls)")
    with torch.cuda.device(x1.device), _mix_scratch(x1.device, Ci, Co, P if window is None else window[0] * window[1], bf16):
        if window is not None:
            fn, size = lib().uno_channel_mix2_win, (*_window_args(window, P, bf16), P)
        else:
            fn, size = (lib().uno_channel_mix2_bf16 if bf16 else lib().uno_channel_mix2), (P,)
        rc = fn(_ptr(x1), _ptr(x2) if x2 is not None else null, C1, _ptr(w), _ptr(bias) if bias is not None else null,
                _ptr(y1), _ptr(y2) if y2 is not None else null, Co1, _ptr(act) if act is not None else null,
                B, Ci, Co, *size, 1 if transpose_w else 0, acc_flag, 1 if act_in else 0,
                _ptr(dgelu_of) if dgelu_of is not None else null,
                _ptr(pw) if pw is not None else null, _ptr(pb) if pb is not None else null, _ptr(proj) if proj is not None else null,
                _stream(x1))
    _check(rc, "uno_channel_mix2")
    if split_out is not None:
        return y1, y2
    if project is not None:
        return y1, proj
    return (y1, act) if y_act else y1


def channel_wgrad_partial_floats(B: int, Ci: int, Co: int, P: int) -> int:
    """floats of split-K partial sums one weight-gradient call of this shape leaves (a whole number of (Co, Ci + 1) blocks)"""
    return int(lib().uno_channel_wgrad_ws_bytes(B, Ci, Co, P)) // 4


def channel_wgrad_finish(parts, Ci: int, Co: int, need_bias: bool, out_w=None, out_b=None, accumulate: bool = False):
    """Second stage alone: `parts` = float32 tensor holding consecutive (Co, Ci + 1) blocks of partial sums (channel_wgrad2(...,
    partials_out=row) for every row) -> gw (Co, Ci), gb (Co) or None, written (accumulate: added) into out_w / out_b when given."""
    _require(parts, torch.float32, "partial sums")
    blk = Co * (Ci + 1)
    if parts.numel() == 0 or parts.numel() % blk:
        raise RuntimeError("uno_amd: the partial sums are not a whole number of (Co, Ci + 1) blocks")
    if out_w is None:
        accumulate = False
        gw = torch.empty((Co, Ci), dtype=torch.float32, device=parts.device)
        gb = torch.empty((Co,), dtype=torch.float32, device=parts.device) if need_bias else None
    else:
        gw, gb = out_w, (out_b if need_bias else None)
        _require(gw, torch.float32, "weight-gradient buffer")
        if gw.numel() != Co * Ci or (need_bias and (gb is None or gb.numel() != Co)):
            raise RuntimeError("uno_amd: gradient buffers do not match the layer")
    with torch.cuda.device(parts.device):
        rc = lib().uno_channel_wgrad_finish(_ptr(parts), _ptr(gw), _ptr(gb) if gb is not None else C.c_void_p(0), Ci, Co,
                                            parts.numel() // blk, 1 if accumulate else 0, _stream(parts))
    _check(rc, "uno_channel_wgrad_finish")
    return gw, gb


def channel_wgrad2(gy, x1, x2, need_bias: bool = True, act_x: bool = False, out_w=None, out_b=None, accumulate: bool = False,
                   partials_out=None, window=None):
    """gy (B, Co, P), x1 (B, C1, P), x2 (B, C2, P) or None -> gw (Co, C1 + C2), gb (Co) or None: the weight gradient of a
    (two-source) layer from one launch (act_x: x1 := gelu(x1) as it is read).  out_w / out_b: write (accumulate: add) into these
    float32 tensors of Co * Ci / Co elements instead of fresh ones (out_b is required with need_bias when out_w is given).
    partials_out: a dense float32 tensor of channel_wgrad_partial_floats() elements - the first stage only, its partial sums left
    there for channel_wgrad_finish (-> None, None).
    window = (rows, cols, pitch): the sums run over that window of the tensors' planes (uno_channel_wgrad2_win, see channel_mix2)."""
    bf16 = _act_dtype(gy, "grad_output")
    _require(x1, gy.dtype, "x1")
    B, Co, P = gy.shape
    if window is not None:
        if partials_out is not None:
            raise RuntimeError("uno_amd: windowed weight gradients run both stages")
        win = _window_args(window, P, bf16)
    C1 = x1.shape[1]
    C2 = 0
    if x2 is not None:
        _require(x2, gy.dtype, "x2")
        C2 = x2.shape[1]
        if x2.shape[0] != B or x2.shape[2] != P:
            raise RuntimeError("uno_amd: grad_output and the sources disagree in batch / pixel count")
    if x1.shape[0] != B or x1.shape[2] != P:
        raise RuntimeError("uno_amd: grad_output and the sources disagree in batch / pixel count")
    Ci = C1 + C2
    L = lib()
    if partials_out is not None:
        _require(partials_out, torch.float32, "partial-sum buffer")
        if partials_out.numel() != channel_wgrad_partial_floats(B, Ci, Co, P):
            raise RuntimeError("uno_amd: partial-sum buffer has the wrong size")
        with torch.cuda.device(gy.device):
            fn = L.uno_channel_wgrad2_bf16 if bf16 else L.uno_channel_wgrad2
            rc = fn(_ptr(gy), _ptr(x1), _ptr(x2) if x2 is not None else C.c_void_p(0), C1, C.c_void_p(0), C.c_void_p(0),
                    _ptr(partials_out), B, Ci, Co, P, 1 if act_x else 0, 3, _stream(gy))
        _check(rc, "uno_channel_wgrad2")
        return None, None
    if out_w is None:
        accumulate = False
        gw = torch.empty((Co, Ci), dtype=torch.float32, device=gy.device)
        gb = torch.empty((Co,), dtype=torch.float32, device=gy.device) if need_bias else None
    else:
        gw, gb = out_w, (out_b if need_bias else None)
        _require(gw, torch.float32, "weight-gradient buffer")
        if gw.numel() != Co * Ci or (need_bias and (gb is None or gb.numel() != Co)):
            raise RuntimeError("uno_amd: gradient buffers do not match the layer")
        if gb is not None:
            _require(gb, torch.float32, "bias-gradient buffer")
    with torch.cuda.device(gy.device):
        if window is not None:
            fn, size, Pl = L.uno_channel_wgrad2_win, (*win, P), win[0] * win[1]
        else:
            fn, size, Pl = (L.uno_channel_wgrad2_bf16 if bf16 else L.uno_channel_wgrad2), (P,), P
        ws = torch.empty(max(1, L.uno_channel_wgrad_ws_bytes(B, Ci, Co, Pl)), dtype=torch.uint8, device=gy.device)
        rc = fn(_ptr(gy), _ptr(x1), _ptr(x2) if x2 is not None else C.c_void_p(0), C1, _ptr(gw), _ptr(gb) if gb is not None else C.c_void_p(0),
                _ptr(ws), B, Ci, Co, *size, 1 if act_x else 0, 1 if accumulate else 0, _stream(gy))
    _check(rc, "uno_channel_wgrad2")
    return gw, gb


def channel_wgrad(gy, x, need_bias: bool = True, act_x: bool = False):
    """gy (B, Co, P), x (B, Ci, P) -> gw (Co, Ci), gb (Co) or None; act_x: x := gelu(x) as it is read."""
    bf16 = _act_dtype(gy, "grad_output")
    _require(x, gy.dtype, "x")
    B, Co, P = gy.shape
    B2, Ci, P2 = x.shape
    if (B2, P2) != (B, P):
        raise RuntimeError("uno_amd: grad_output and x disagree in batch / pixel count")
    L = lib()
    gw = torch.empty((Co, Ci), dtype=torch.float32, device=x.device)
    gb = torch.empty((Co,), dtype=torch.float32, device=x.device) if need_bias else None
    with torch.cuda.device(x.device):
        ws = torch.empty(max(1, L.uno_channel_wgrad_ws_bytes(B, Ci, Co, P)), dtype=torch.uint8, device=x.device)
        fn = L.uno_channel_wgrad_bf16 if bf16 else L.uno_channel_wgrad
        rc = fn(_ptr(gy), _ptr(x), _ptr(gw), _ptr(gb) if need_bias else C.c_void_p(0), _ptr(ws),
                                 B, Ci, Co, P, 1 if act_x else 0, _stream(x))
    _check(rc, "uno_channel_wgrad")
    return gw, gb


def adam_step(p, g, m, v, step: int, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float):
    """In-place Adam update of one parameter tensor p (float32 or complex64) with gradient g, first moment m (real
    view shape) and second moment v (p.shape, real)."""
    cplx = p.is_complex()
    pr, gr = (torch.view_as_real(p), torch.view_as_real(g)) if cplx else (p, g)
    for t, name in ((pr, "param"), (gr, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
        _require(t, torch.float32, name)
    if gr.numel() != pr.numel() or m.numel() != pr.numel() or v.numel() != p.numel():
        raise RuntimeError("uno_amd: Adam state shapes do not match the parameter")
    with torch.cuda.device(p.device):
        rc = lib().uno_adam_step(_ptr(pr), _ptr(gr), _ptr(m), _ptr(v), p.numel(), 1 if cplx else 0, lr, beta1, beta2, eps,
                                 weight_decay, int(step), _stream(pr))
    _check(rc, "uno_adam_step")


def gelu_project_forward(pre, w, bias=None):
    """pre (B, C, P) f32, w (C,), bias (1,) or None -> out (B, P) = bias + sum_c w[c] gelu(pre[:, c])."""
    bf16 = _act_dtype(pre, "pre")
    _require(w, torch.float32, "weight")
    if bias is not None:
        _require(bias, torch.float32, "bias")
    B, Cc, P = pre.shape
    if w.numel() != Cc:
        raise RuntimeError(f"uno_amd: weight has {w.numel()} entries for {Cc} channels")
    out = torch.empty((B, P), dtype=pre.dtype, device=pre.device)
    with torch.cuda.device(pre.device):
        rc = (lib().uno_gelu_project_forward_bf16 if bf16 else lib().uno_gelu_project_forward)(_ptr(pre), _ptr(w), _ptr(bias) if bias is not None else C.c_void_p(0), _ptr(out),
                                            B, Cc, P, _stream(pre))
    _check(rc, "uno_gelu_project_forward")
    return out


def gelu_project_backward(pre, w, gout, need_bias=True, window=None):
    """-> gpre (B, C, P), gw (C,), gb (1,) or None.  window = (rows, cols, pitch): on that window of the planes (see channel_mix2)."""
    bf16 = _act_dtype(pre, "pre")
    _require(w, torch.float32, "weight")
    _require(gout, pre.dtype, "grad_output")
    B, Cc, P = pre.shape
    if tuple(gout.shape) != (B, P):
        raise RuntimeError("uno_amd: grad_output shape does not match (batch, pixels)")
    L = lib()
    gpre = torch.empty_like(pre)
    gw = torch.empty((Cc,), dtype=torch.float32, device=pre.device)
    gb = torch.empty((1,), dtype=torch.float32, device=pre.device) if need_bias else None
    with torch.cuda.device(pre.device):
        if window is not None:
            win = _window_args(window, P, bf16)
            fn, size, Pl = L.uno_gelu_project_backward_win, (*win, P), win[0] * win[1]
        else:
            fn, size, Pl = (L.uno_gelu_project_backward_bf16 if bf16 else L.uno_gelu_project_backward), (P,), P
        ws = torch.empty(max(1, L.uno_gelu_project_bwd_ws_bytes(B, Cc, Pl)), dtype=torch.uint8, device=pre.device)
        rc = fn(_ptr(pre), _ptr(w), _ptr(gout), _ptr(gpre), _ptr(gw), _ptr(gb) if need_bias else C.c_void_p(0), _ptr(ws), B, Cc, *size,
                _stream(pre))
    _check(rc, "uno_gelu_project_backward")
    return gpre, gw, gb


class _DeviceView:
    """a library-owned device allocation seen through the CUDA array interface"""
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


def table_to_device(t: torch.Tensor, device) -> torch.Tensor:
    """A host-built operand table on the device.  Outside a graph capture: `t.to(device)`.  While the current stream is being captured
    (a shape first seen inside a capture) the copy torch would make is not permitted: the table then goes through uno_upload_table
    (an allocation of the library's own, uploaded under the relaxed capture mode, never freed - the callers cache per shape)."""
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_current_stream_capturing():
        return t.to(device)
    t = t.contiguous()
    if t.numel() == 0:
        return torch.empty(t.shape, dtype=t.dtype, device=device)
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.int64: "<i8", torch.float64: "<f8"}[t.dtype]
    with torch.cuda.device(device):
        ptr = lib().uno_upload_table(C.c_void_p(t.data_ptr()), t.numel() * t.element_size())
    if not ptr:
        _check(-5, "uno_upload_table")
    return torch.as_tensor(_DeviceView(ptr, t.shape, typestr), device=device)


def _project_geometry(window, P: int):
    if window is None:
        return (0, 0, 0, P), P
    rows, cols, pitch = _window_args(window, P, False)
    return (rows, cols, pitch, P), rows * cols


def project_backward_applies(B: int, C1: int, Ci: int, Co: int, P: int, window=None) -> bool:
    """Does uno_project_backward take fc1 (Ci -> Co, sources split at C1; C1 = Ci: one source) on planes of P elements [on that window]?"""
    if window is not None:
        rows, cols, pitch = (int(v) for v in window)
        if rows < 1 or cols % 4 or cols < 260 or pitch < cols or (rows - 1) * pitch + cols > P or rows * cols >= 1 << 24:
            return False
        geo = (rows, cols, pitch, P)
    else:
        geo = (0, 0, 0, P)
    return bool(lib().uno_project_backward_applies(B, C1, Ci, Co, *geo))


def project_backward(x1, x2, w, pre, w2, gout, act_in: bool = False, need_bias: bool = True, need_bias2: bool = True, window=None,
                     out_w=None, out_b=None, accumulate: bool = False):
    """The backward pass of `fc2(gelu(fc1(cat([gelu](x1), x2))))` (uno_project_backward, ABI 12): x1 (B, C1, P), x2 (B, C2, P) or None,
    w (Co, C1 + C2), pre (B, Co, P) = fc1's output, w2 (Co), gout (B, P) -> g1, g2 (or None), gw (Co, Ci), gb (Co) or None, gw2 (Co),
    gb2 (1) or None.  The gradient at fc1's output is formed inside the two kernels.  window: see channel_mix2 (elements outside the
    window of the fresh g1 / g2 are left as they are).  out_w / out_b / accumulate: as channel_wgrad2."""
    for t, name in ((x1, "x1"), (w, "weight"), (pre, "pre"), (w2, "weight2"), (gout, "grad_output")):
        _require(t, torch.float32, name)
    B, C1, P = x1.shape
    C2 = 0
    if x2 is not None:
        _require(x2, torch.float32, "x2")
        C2 = x2.shape[1]
        if x2.shape[0] != B or x2.shape[2] != P:
            raise RuntimeError("uno_amd: the two sources disagree in batch / pixel count")
    Ci, Co = C1 + C2, pre.shape[1]
    if tuple(pre.shape) != (B, Co, P) or tuple(gout.shape) != (B, P) or w.numel() != Co * Ci or w2.numel() != Co:
        raise RuntimeError("uno_amd: project_backward: operand shapes do not match the two layers")
    geo, Pl = _project_geometry(window, P)
    L = lib()
    g1 = torch.empty_like(x1)
    g2 = torch.empty_like(x2) if x2 is not None else None
    if out_w is None:
        accumulate = False
        gw = torch.empty((Co, Ci), dtype=torch.float32, device=x1.device)
        gb = torch.empty((Co,), dtype=torch.float32, device=x1.device) if need_bias else None
    else:
        gw, gb = out_w, (out_b if need_bias else None)
        _require(gw, torch.float32, "weight-gradient buffer")
        if gw.numel() != Co * Ci or (need_bias and (gb is None or gb.numel() != Co)):
            raise RuntimeError("uno_amd: gradient buffers do not match the layer")
        if gb is not None:
            _require(gb, torch.float32, "bias-gradient buffer")
    gw2 = torch.empty((Co,), dtype=torch.float32, device=x1.device)
    gb2 = torch.empty((1,), dtype=torch.float32, device=x1.device) if need_bias2 else None
    null = C.c_void_p(0)
    with torch.cuda.device(x1.device):
        ws = torch.empty(max(1, L.uno_project_backward_ws_bytes(B, Ci, Co, Pl)), dtype=torch.uint8, device=x1.device)
        rc = L.uno_project_backward(_ptr(x1), _ptr(x2) if x2 is not None else null, C1, _ptr(w), _ptr(pre), _ptr(w2), _ptr(gout),
                                    _ptr(g1), _ptr(g2) if g2 is not None else null, _ptr(gw), _ptr(gb) if gb is not None else null,
                                    _ptr(gw2), _ptr(gb2) if gb2 is not None else null, _ptr(ws), B, Ci, Co, *geo,
                                    1 if act_in else 0, 1 if accumulate else 0, _stream(x1))
    _check(rc, "uno_project_backward")
    return g1, g2, gw, gb, gw2, gb2


def gelu_pad(s, Hp: int, Wp: int):
    """s (..., H, W) f32 -> (..., Hp, Wp) = zero-pad(gelu(s)) at the end of both axes."""
    bf16 = _act_dtype(s, "s")
    *lead, H, W = s.shape
    n = 1
    for d in lead:
        n *= d
    out = torch.empty((*lead, Hp, Wp), dtype=s.dtype, device=s.device)
    with torch.cuda.device(s.device):
        rc = (lib().uno_gelu_pad_bf16 if bf16 else lib().uno_gelu_pad)(_ptr(s), C.c_void_p(0), _ptr(out), n, H, W, Hp, Wp, 0, _stream(s))
    _check(rc, "uno_gelu_pad")
    return out


def gelu_pad_backward(s, gy):
    """gs (..., H, W) = gelu'(s) * gy[..., :H, :W]."""
    bf16 = _act_dtype(s, "s")
    _require(gy, s.dtype, "grad_output")
    *lead, H, W = s.shape
    Hp, Wp = gy.shape[-2:]
    n = 1
    for d in lead:
        n *= d
    out = torch.empty_like(s)
    with torch.cuda.device(s.device):
        rc = (lib().uno_gelu_pad_bf16 if bf16 else lib().uno_gelu_pad)(_ptr(s), _ptr(gy), _ptr(out), n, H, W, Hp, Wp, 1, _stream(s))
    _check(rc, "uno_gelu_pad")
    return out


def channels_last_pitch(t):
    """(pitch, batch stride) in elements if `t` (B, C, *grid) is stored channels-LAST - unit stride over the channels, the grid
    points dense at one common pitch >= C (a channel slice of a wider channels-last tensor qualifies), any batch stride - else None."""
    if t.dim() < 3 or t.shape[1] < 2 or t.stride(1) != 1:
        return None
    ld = t.stride(-1)
    if ld < t.shape[1]:
        return None
    run = ld
    for d in range(t.dim() - 1, 1, -1):
        if t.shape[d] != 1 and t.stride(d) != run:
            return None
        run *= t.shape[d]
    if t.shape[0] > 1 and t.stride(0) < (run // ld - 1) * ld + t.shape[1]:
        return None
    return ld, t.stride(0)


def to_channels_first(t):
    """A channels-last f32 activation (see channels_last_pitch) as a contiguous (B, C, *grid) tensor, by the tiled transposing copy."""
    _require_dev(t, torch.float32, "activation")
    ld, sb = channels_last_pitch(t)
    B, Cc = t.shape[:2]
    P = 1
    for d in t.shape[2:]:
        P *= d
    out = torch.empty(t.shape, dtype=t.dtype, device=t.device)
    with torch.cuda.device(t.device):
        rc = lib().uno_transpose_batched(_ptr(t), _ptr(out), B, P, Cc, ld, sb if B > 1 else 0, P, Cc * P, _stream(t))
    _check(rc, "uno_transpose_batched")
    return out


def instnorm_forward(x, gamma, beta, eps: float, gelu: bool):
    """x (B, C, *grid) f32 -> y, mean (B*C), rstd (B*C)."""
    bf16 = _act_dtype(x, "x")
    for t, name in ((gamma, "weight"), (beta, "bias")):
        if t is not None:
            _require(t, torch.float32, name)
    B, Cc = x.shape[0], x.shape[1]
    rows = B * Cc
    N = x.numel() // max(rows, 1)
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    null = C.c_void_p(0)
    with torch.cuda.device(x.device):
        rc = (lib().uno_instnorm_forward_bf16 if bf16 else lib().uno_instnorm_forward)(_ptr(x), _ptr(gamma) if gamma is not None else null, _ptr(beta) if beta is not None else null,
                                        _ptr(y), _ptr(mean), _ptr(rstd), rows, Cc, N, float(eps), 1 if gelu else 0, _stream(x))
    _check(rc, "uno_instnorm_forward")
    return y, mean, rstd


def instnorm_backward(x, gy, gamma, beta, mean, rstd, gelu: bool):
    """-> gx, s1 (B, C), s2 (B, C): sums over the batch of s1 / s2 are the bias / weight gradients."""
    bf16 = _act_dtype(x, "x")
    _require(gy, x.dtype, "grad_output")
    B, Cc = x.shape[0], x.shape[1]
    rows = B * Cc
    N = x.numel() // max(rows, 1)
    gx = torch.empty_like(x)
    s1 = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    s2 = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    null = C.c_void_p(0)
    with torch.cuda.device(x.device):
        rc = (lib().uno_instnorm_backward_bf16 if bf16 else lib().uno_instnorm_backward)(_ptr(x), _ptr(gy), _ptr(gamma) if gamma is not None else null,
                                         _ptr(beta) if beta is not None else null, _ptr(mean), _ptr(rstd), _ptr(gx), _ptr(s1), _ptr(s2),
                                         rows, Cc, N, 1 if gelu else 0, _stream(x))
    _check(rc, "uno_instnorm_backward")
    return gx, s1, s2


class AdamPlan:
    """Pointer tables of a fixed set of parameter tensors (params, grads, moments must not be re-allocated): one
    native call per optimiser step."""

    def __init__(self, params, grads, ms, vs):
        self.device = params[0].device
        self.n = len(params)
        reals = []
        for p, g, m, v in zip(params, grads, ms, vs):
            cplx = p.is_complex()
            pr, gr = (torch.view_as_real(p), torch.view_as_real(g)) if cplx else (p, g)
            for t, name in ((pr, "param"), (gr, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
                _require(t, torch.float32, name)
            if gr.numel() != pr.numel() or m.numel() != pr.numel() or v.numel() != p.numel():
                raise RuntimeError("uno_amd: Adam state shapes do not match the parameter")
            reals.append((pr, gr, m, v, p.numel(), 1 if cplx else 0))
        # parameters and moments are kept alive; gradient buffers are NOT (a plan must not pin the gradients of a step that
        # released them, e.g. zero_grad(set_to_none=True)): the caller re-validates the pointer tuple (`key`) before every use
        self._keep = [(r[0], r[2], r[3]) for r in reals]
        arr = _fp * self.n
        self.p = arr(*[r[0].data_ptr() for r in reals])
        self.g = arr(*[r[1].data_ptr() for r in reals])
        self.m = arr(*[r[2].data_ptr() for r in reals])
        self.v = arr(*[r[3].data_ptr() for r in reals])
        self.sizes = (C.c_longlong * self.n)(*[r[4] for r in reals])
        self.cplx = (_i * self.n)(*[r[5] for r in reals])
        self.key = tuple((r[0].data_ptr(), r[1].data_ptr(), r[2].data_ptr(), r[3].data_ptr()) for r in reals)

    def step(self, step: int, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float):
        with torch.cuda.device(self.device):
            rc = lib().uno_adam_step_multi(self.n, self.p, self.g, self.m, self.v, self.sizes, self.cplx, lr, beta1, beta2, eps,
                                           weight_decay, int(step), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _check(rc, "uno_adam_step_multi")

    def step_dev(self, counter, scalars, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, hyper=None):
        """the update with the step count on the device (`counter`: int32 tensor of one element, advanced here; `scalars`: four floats
        of scratch; `hyper`: None or a float64 tensor (lr, eps, weight_decay) read at execution time instead of the arguments):
        capturable in a HIP graph"""
        if scalars.numel() < 4 or scalars.dtype != torch.float32 or counter.dtype != torch.int32:
            raise RuntimeError("uno_amd: Adam device scratch must be an int32 counter and four float32 scalars")
        if hyper is not None and (hyper.dtype != torch.float64 or hyper.numel() < 3 or hyper.device != self.device):
            raise RuntimeError("uno_amd: Adam device hyper-parameters must be three float64 values on the parameters' device")
        with torch.cuda.device(self.device):
            rc = lib().uno_adam_step_multi_dev(self.n, self.p, self.g, self.m, self.v, self.sizes, self.cplx, lr, beta1, beta2, eps,
                                               weight_decay, _ptr(counter), _ptr(scalars), _ptr(hyper) if hyper is not None else None,
                                               C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _check(rc, "uno_adam_step_multi_dev")


def profile_begin(max_records: int = 100000):
    _check(lib().uno_profile_begin(int(max_records)), "uno_profile_begin")


def profile_end():
    """-> list of (kernel name, milliseconds, algorithmic bytes), one per kernel launch recorded."""
    L = lib()
    n = L.uno_profile_end()
    out = []
    name = C.create_string_buffer(64)
    ms, by = C.c_double(), C.c_double()
    for i in range(n):
        _check(L.uno_profile_get(i, name, 64, C.byref(ms), C.byref(by)), "uno_profile_get")
        out.append((name.value.decode(), ms.value, by.value))
    return out
