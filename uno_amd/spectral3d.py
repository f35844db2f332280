"""3-D spectral convolution on the HIP path (SpectralConv3d_Uno.forward, reference
integral_operators.py:385-427): rfftn over (H, W, T) restricted to the four low-frequency corners ->
per-mode channel mixing with weights1..4 -> zero-padded irfftn, with a custom autograd adjoint that
saves only the truncated input spectrum."""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _native


def _plain(t):
    if t.is_complex() and t.is_conj():
        t = t.resolve_conj()
    if t.is_neg():
        t = t.resolve_neg()
    return t.contiguous()


class _SpectralConv3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, w2, w3, w4, d1, d2, d3):
        x = _plain(x)
        ws = [_plain(w) for w in (w1, w2, w3, w4)]
        y, xt = _native.spectral_conv3d_forward(x, ws, int(d1), int(d2), int(d3))
        ctx.save_for_backward(xt, *ws)
        ctx.in_dims = tuple(x.shape[-3:])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xt, *ws = ctx.saved_tensors
        need_gx = ctx.needs_input_grad[0]
        need_gw = any(ctx.needs_input_grad[1:5])
        gx, gws = _native.spectral_conv3d_backward(_plain(gy), xt, ws, *ctx.in_dims, need_gx=need_gx, need_gw=need_gw)
        gws = gws or [None] * 4
        return (gx, *gws, None, None, None)


def spectral_conv3d(x, weights, dim1, dim2, dim3):
    return _SpectralConv3dFn.apply(x, *weights, dim1, dim2, dim3)
