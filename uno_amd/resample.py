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
def _matrix(n_in: int, n_out: int, mode: str = "bicubic_aa") -> torch.Tensor:
    """1-D operator (n_out, n_in) of torch's own float32 CPU op: "bicubic_aa" = bicubic / antialias / align_corners (the 2-D
    point-wise branch), "linear" = the per-axis factor of trilinear / align_corners (skip connections of the 3-D model,
    reference navier_stokes_uno3d.py:352-372)."""
    # a second axis of size 2 -> 2 is an exact identity under align_corners, and avoids torch's
    # degenerate handling of a length-1 axis
    eye = torch.eye(n_in, dtype=torch.float32).view(1, n_in, n_in, 1).expand(1, n_in, n_in, 2).contiguous()
    if mode == "bicubic_aa":
        r = F.interpolate(eye, size=(n_out, 2), mode="bicubic", align_corners=True, antialias=True)
    elif mode == "linear":
        r = F.interpolate(eye, size=(n_out, 2), mode="bilinear", align_corners=True)
    else:
        raise ValueError(mode)
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
def _tables(n_in: int, n_out: int, device_str: str, mode: str = "bicubic_aa"):
    """per direction (forward R, adjoint R^T): ((start, weights) band table, (p0, dense tile weights)) on the device."""
    R = _matrix(n_in, n_out, mode)
    dev = torch.device(device_str)
    out = []
    for M in (R, R.t().contiguous()):
        s, w, _ = _band(M)
        p0, tw = _row_tiles(M)
        up = lambda t: _native.table_to_device(t.contiguous(), dev)          # (also inside a graph capture)
        out.append(((up(s), up(w)), (up(p0), up(tw))))
    return tuple(out)


ADD_KE = 3          # k-steps of the fused up-sampling operands (csrc/dft2d_inv_add_kernel.h): bands of <= 4 * 3 = 12 sources per 16 outputs


def _add_operands(Rh: torch.Tensor, Rw: torch.Tensor):
    """Operand tables of uno_dft2d_inverse_add for the separable operator out = Rh . t . Rw^T (Rh (H, Hs), Rw (W, Ws) dense float32), or
    None when a band is too wide.  Layouts: include/uno_spectral.h."""
    H, Hs = Rh.shape
    W, Ws = Rw.shape
    span = 4 * ADD_KE
    if Hs < 1 or Ws < span:
        return None
    lane = torch.arange(64)
    r16, kk = lane % 16, lane // 16

    def band(M, rows):
        """(first, last) non-zero column over the given rows of M (rows outside the matrix are skipped); (0, -1) when all are zero"""
        rows = [r for r in rows if 0 <= r < M.shape[0]]
        if not rows:
            return 0, -1
        nz = (M[rows] != 0).any(0).nonzero().flatten()
        return (int(nz[0]), int(nz[-1])) if nz.numel() else (0, -1)

    nrt = (H + 15) // 16
    p0 = torch.zeros(nrt, dtype=torch.int32)
    rowop = torch.zeros(nrt, ADD_KE, 64, dtype=torch.float32)
    Rh_pad = torch.zeros(16 * nrt, Hs + span, dtype=torch.float32)
    Rh_pad[:H, :Hs] = Rh
    for rt in range(nrt):
        lo, hi = band(Rh, range(16 * rt, 16 * rt + 16))
        if hi - lo + 1 > span:
            return None
        p0[rt] = lo
        for e in range(ADD_KE):
            rowop[rt, e] = Rh_pad[16 * rt + r16, lo + ADD_KE * kk + e]
    nwt = ((W // 2) + 16) // 16
    v0 = torch.zeros(nwt, 2, dtype=torch.int32)
    colop = torch.zeros(nwt, 2, ADD_KE, 64, dtype=torch.float32)
    Rw_pad = torch.zeros(W + 1, Ws + span, dtype=torch.float32)      # row W: all zero (columns outside [0, W))
    Rw_pad[:W, :Ws] = Rw
    for wt in range(nwt):
        for side in range(2):
            cols = [16 * wt + i if side == 0 else W - 16 * wt - i for i in range(16)]
            lo, hi = band(Rw, cols)
            lo = max(min(lo, Ws - span), 0)          # the 12-byte pieces of a lane stay inside the source row
            if hi - lo + 1 > span:
                return None
            v0[wt, side] = lo
            w = torch.tensor([c if 0 <= c < W else W for c in cols])[r16]
            for ks in range(ADD_KE):
                colop[wt, side, ks] = Rw_pad[w, lo + ADD_KE * kk + ks]
    return p0, rowop.contiguous(), v0.contiguous(), colop.contiguous()


@functools.lru_cache(maxsize=None)
def upsample_add_tables(Hs: int, Ws: int, H: int, W: int, device_str: str, adjoint: bool = False, mode: str = "bicubic_aa"):
    """Device operand tables for the fused `inverse transform + resampled addend` kernel, or None.
    adjoint=False: the addend is resample_forward of an (Hs, Ws) image to (H, W) (an up-sampling block's point-wise branch);
    adjoint=True: the addend is resample_adjoint, for an (H, W) INPUT grid, of an (Hs, Ws) gradient (a down-sampling block's input
    gradient: the operators are the transposes of the (H, W) -> (Hs, Ws) matrices)."""
    if adjoint:
        Rh, Rw = _matrix(H, Hs, mode).t().contiguous(), _matrix(W, Ws, mode).t().contiguous()
    else:
        Rh, Rw = _matrix(Hs, H, mode), _matrix(Ws, W, mode)
    ops = _add_operands(Rh, Rw)
    if ops is None:
        return None
    dev = torch.device(device_str)
    return tuple(_native.table_to_device(t, dev) for t in ops)


def resample_forward(x: torch.Tensor, Ho: int, Wo: int, out: torch.Tensor | None = None, reverse: bool = False) -> torch.Tensor:
    """R_h x R_w^T on the device (no autograd); with `out`, accumulates into it.  reverse: images in descending order (same result)."""
    H, W = x.shape[-2:]
    (fh, th), _ = _tables(H, Ho, str(x.device))
    (fw, _), _ = _tables(W, Wo, str(x.device))
    return _native.resample2d(x, Ho, Wo, fh, fw, th, out=out, reverse=reverse)


def resample_adjoint(gy: torch.Tensor, H: int, W: int, out: torch.Tensor | None = None, reverse: bool = False) -> torch.Tensor:
    """Adjoint of resample_forward for an (H, W) input grid: R_h^T gy R_w; with `out`, accumulates into it."""
    Ho, Wo = gy.shape[-2:]
    _, (bh, th) = _tables(H, Ho, str(gy.device))
    _, (bw, _) = _tables(W, Wo, str(gy.device))
    return _native.resample2d(gy, H, W, bh, bw, th, out=out, reverse=reverse)


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


# ------------------------------------------------------------------------------------------------ trilinear (3-D skips)
def _apply3d(x: torch.Tensor, sizes, adjoint: bool) -> torch.Tensor:
    """Separable trilinear / align_corners operator on (B, C, D1, D2, D3) with the banded kernels: the last two axes in one
    fused pass over (B*C*D1) images, the first axis as the row operator of (B*C) images of D2*D3 columns (identity on the
    columns).  adjoint: x is the gradient on the `sizes`-shaped... see callers for the argument order."""
    dev = str(x.device)
    B, C = x.shape[:2]
    (i1, i2, i3), (o1, o2, o3) = sizes              # operator maps (i1, i2, i3) -> (o1, o2, o3); the adjoint maps back
    k = 1 if adjoint else 0
    src, dst = ((o1, o2, o3), (i1, i2, i3)) if adjoint else ((i1, i2, i3), (o1, o2, o3))

    def tabs(n_in, n_out):
        return _tables(n_in, n_out, dev, "linear")[k]       # (band table, row tiles) of R or R^T

    def last_two(t, d1):
        if (src[1], src[2]) == (dst[1], dst[2]):
            return t
        (bh, th), (bw, _) = tabs(i2, o2), tabs(i3, o3)
        return _native.resample2d(t.reshape(B * C * d1, src[1], src[2]), dst[1], dst[2], bh, bw, th).view(B, C, d1, dst[1], dst[2])

    def first(t, d2, d3):
        if src[0] == dst[0]:
            return t
        (bh, th) = tabs(i1, o1)
        (bw, _) = _tables(d2 * d3, d2 * d3, dev, "linear")[0]          # identity on the flattened (D2, D3) columns
        return _native.resample2d(t.reshape(B * C, src[0], d2 * d3), dst[0], d2 * d3, bh, bw, th).view(B, C, dst[0], d2, d3)

    # run the shrinking step first: less data through the second one
    if src[0] * dst[1] * dst[2] <= dst[0] * src[1] * src[2]:
        return first(last_two(x, src[0]), dst[1], dst[2])
    return last_two(first(x, src[1], src[2]), dst[0])


class _Trilinear3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size):
        x = x.contiguous()
        ctx.sizes = (tuple(x.shape[2:]), tuple(size))
        return _apply3d(x, ctx.sizes, adjoint=False)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return _apply3d(gy.contiguous(), ctx.sizes, adjoint=True), None


def resample3d_trilinear(x: torch.Tensor, size) -> torch.Tensor:
    """== F.interpolate(x, size=size, mode="trilinear", align_corners=True) for 5-D x; float32 device tensors run the banded
    kernels with torch's own float32 weights (the stock backward is an atomics kernel: 20 ms of a 60 ms NS-3D step)."""
    size = tuple(int(v) for v in size)
    if tuple(x.shape[2:]) == size:
        return x                                    # exact identity under align_corners
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and min(x.shape[2:]) > 1 and min(size) > 1:
        return _Trilinear3dFn.apply(x, size)
    return F.interpolate(x, size=size, mode="trilinear", align_corners=True)
