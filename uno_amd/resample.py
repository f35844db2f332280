"""Bicubic anti-aliased (align_corners=True) resampling of pointwise_op_2D (reference
integral_operators.py:240-242) as a separable banded operator on the HIP path.

The 1-D resampling matrix R (out x in) for a size pair is read off torch's own CPU op once (float32 - the
reference computes its weights in float32, whose rounding of scale * index is visible at the 1e-5 level on
446-point axes - identity input with a dummy pass-through axis), so its weights are exactly what the reference applies; it is stored as a
band table (first column, K taps per row) together with the band table of R^T for the adjoint.  The device
kernels (csrc/resample2d.hip) apply the two 1-D operators; autograd uses the transposed tables.
"""
from __future__ import annotations

import functools

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _native


@functools.lru_cache(maxsize=None)
def _matrix(n_in: int, n_out: int) -> torch.Tensor:
    # a second axis of size 2 -> 2 is an exact identity under align_corners, and avoids torch's
    # degenerate handling of a length-1 axis
    eye = torch.eye(n_in, dtype=torch.float32).view(1, n_in, n_in, 1).expand(1, n_in, n_in, 2).contiguous()
    r = F.interpolate(eye, size=(n_out, 2), mode="bicubic", align_corners=True, antialias=True)
    return r[0, :, :, 0].t().contiguous()          # (n_out, n_in)


def _band(mat: torch.Tensor):
    nz = mat != 0
    n_out, n_in = mat.shape
    first = torch.where(nz.any(1), nz.float().argmax(1), torch.zeros(n_out, dtype=torch.long))
    last = torch.where(nz.any(1), n_in - 1 - nz.flip(1).float().argmax(1), torch.zeros(n_out, dtype=torch.long))
    K = int((last - first + 1).max())
    cols = first[:, None] + torch.arange(K)[None, :]
    w = torch.where(cols < n_in, mat.gather(1, cols.clamp(max=n_in - 1)), torch.zeros((), dtype=mat.dtype))
    return first.to(torch.int32), w.to(torch.float32), K


@functools.lru_cache(maxsize=None)
def _tables(n_in: int, n_out: int, device_str: str):
    """((start, weights) of R, (start, weights) of R^T) on the device."""
    R = _matrix(n_in, n_out)
    dev = torch.device(device_str)
    s, w, _ = _band(R)
    st, wt, _ = _band(R.t().contiguous())
    return (s.to(dev), w.contiguous().to(dev)), (st.to(dev), wt.contiguous().to(dev))


class _Resample2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x = x.contiguous()
        H, W = x.shape[-2:]
        ctx.sizes = (H, W, Ho, Wo)
        fh, _ = _tables(H, Ho, str(x.device))
        fw, _ = _tables(W, Wo, str(x.device))
        return _native.resample2d(x, Ho, Wo, fh, fw)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        H, W, Ho, Wo = ctx.sizes
        _, bh = _tables(H, Ho, str(gy.device))
        _, bw = _tables(W, Wo, str(gy.device))
        return _native.resample2d(gy.contiguous(), H, W, bh, bw), None, None


def resample2d_bicubic_aa(x: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """== F.interpolate(x, size=(Ho, Wo), mode="bicubic", align_corners=True, antialias=True) for 4-D float32 x."""
    if x.shape[-2] == Ho and x.shape[-1] == Wo:
        return x            # the operator is the identity for equal sizes (weights 0, 1, 0, 0)
    return _Resample2dFn.apply(x, int(Ho), int(Wo))
