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


TILE_ROWS = 16


def _row_tiles(mat: torch.Tensor):
    """Dense 16-output-row tiles of a banded operator: (first input row p0 [ntiles], weights [ntiles, NP, 16])
    with weights[k, u, r] = mat[16 k + r, p0[k] + u] - the form the fused kernel's row phase consumes."""
    n_out, n_in = mat.shape
    first, w, K = _band(mat)
    ntiles = (n_out + TILE_ROWS - 1) // TILE_ROWS
    p0 = torch.zeros(ntiles, dtype=torch.int32)
    spans = []
    for k in range(ntiles):
        rows = range(TILE_ROWS * k, min(TILE_ROWS * (k + 1), n_out))
        lo = min(int(first[i]) for i in rows)
        hi = max(min(int(first[i]) + K, n_in) for i in rows)
        p0[k] = lo
        spans.append(hi - lo)
    NP = max(spans)
    tw = torch.zeros(ntiles, NP, TILE_ROWS, dtype=torch.float32)
    for k in range(ntiles):
        for r in range(TILE_ROWS):
            i = TILE_ROWS * k + r
            if i >= n_out:
                break
            lo = int(p0[k])
            seg = mat[i, lo:min(lo + NP, n_in)].to(torch.float32)
            tw[k, :seg.numel(), r] = seg
    return p0, tw


@functools.lru_cache(maxsize=None)
def _tables(n_in: int, n_out: int, device_str: str):
    """per direction (forward R, adjoint R^T): ((start, weights) band table, (p0, dense tile weights)) on the device."""
    R = _matrix(n_in, n_out)
    dev = torch.device(device_str)
    out = []
    for M in (R, R.t().contiguous()):
        s, w, _ = _band(M)
        p0, tw = _row_tiles(M)
        out.append(((s.to(dev), w.contiguous().to(dev)), (p0.to(dev), tw.contiguous().to(dev))))
    return tuple(out)


def resample_forward(x: torch.Tensor, Ho: int, Wo: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """R_h x R_w^T on the device (no autograd); with `out`, accumulates into it."""
    H, W = x.shape[-2:]
    (fh, th), _ = _tables(H, Ho, str(x.device))
    (fw, _), _ = _tables(W, Wo, str(x.device))
    return _native.resample2d(x, Ho, Wo, fh, fw, th, out=out)


def resample_adjoint(gy: torch.Tensor, H: int, W: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """Adjoint of resample_forward for an (H, W) input grid: R_h^T gy R_w; with `out`, accumulates into it."""
    Ho, Wo = gy.shape[-2:]
    _, (bh, th) = _tables(H, Ho, str(gy.device))
    _, (bw, _) = _tables(W, Wo, str(gy.device))
    return _native.resample2d(gy, H, W, bh, bw, th, out=out)


class _Resample2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x = x.contiguous()
        ctx.in_hw = tuple(x.shape[-2:])
        return resample_forward(x, Ho, Wo)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return resample_adjoint(gy.contiguous(), *ctx.in_hw), None, None


def resample2d_bicubic_aa(x: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """== F.interpolate(x, size=(Ho, Wo), mode="bicubic", align_corners=True, antialias=True) for 4-D float32 x."""
    if x.shape[-2] == Ho and x.shape[-1] == Wo:
        return x            # the operator is the identity for equal sizes (weights 0, 1, 0, 0)
    return _Resample2dFn.apply(x, int(Ho), int(Wo))
