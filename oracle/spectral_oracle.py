"""CPU oracle for the U-NO spectral-convolution hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  Nothing under ``uno_amd/`` imports it and
the product path never routes through it.

Parity status: PINNED.  The reference (ashiq24/UNO) has no tests of its own, so
the oracle is pinned by golden vectors produced by importing the genuine
reference in the build container (``oracle/gen_golden.py`` -> ``tests/golden``);
``tests/test_oracle_golden.py`` checks every function below against them.

Two independent restatements of the same operator are kept:

* ``*_dense``  - closed form with dense truncated DFT matrices in float64
  (no FFT library involved).  This is the mathematical contract the HIP kernels
  implement (pruned forward DFT -> per-mode complex GEMM -> pruned inverse DFT).
* ``*_fft``    - the op sequence the reference executes
  (rfft2/rfftn -> einsum on the low-frequency corners -> irfft2/irfftn), kept
  differentiable so it can serve as the timed CPU baseline.

Reference lines followed (relative to the reference checkout):
  integral_operators.py:181-207  SpectralConv2d_Uno.forward
  integral_operators.py:385-427  SpectralConv3d_Uno.forward
  integral_operators.py:224-243  pointwise_op_2D.forward
  integral_operators.py:440-468  pointwise_op_3D.forward
  integral_operators.py:272-284, 500-513  OperatorBlock_2D/3D.forward
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# index bookkeeping shared by the dense forms
# --------------------------------------------------------------------------- #
def corner_rows(n: int, m: int) -> np.ndarray:
    """Spectrum rows touched by the two corners of one full-complex axis.

    integral_operators.py:198-203 - rows ``[:m]`` then rows ``[-m:]`` of an axis
    of length ``n``; returned in weight order (corner "lo" first, then "hi").
    """
    return np.concatenate([np.arange(m), np.arange(n - m, n)])


def later_wins_mask(n_out: int, m: int) -> np.ndarray:
    """1 where a corner-row entry survives in ``out_ft``, 0 where a later
    slice-assignment overwrote it (integral_operators.py:198-203 ordering).

    Entry j < m (lo corner, row j) is overwritten when j >= n_out - m.
    """
    keep = np.ones(2 * m)
    lo = np.arange(m)
    keep[:m] = (lo < n_out - m).astype(np.float64)
    return keep


def hermitian_weights(n_out: int, m: int) -> np.ndarray:
    """c_l of the one-sided inverse (irfft) along the half-spectrum axis:
    1 for l = 0 and for the Nyquist bin of an even axis, else 2."""
    c = np.full(m, 2.0)
    c[0] = 1.0
    if n_out % 2 == 0 and m - 1 == n_out // 2:
        c[-1] = 1.0
    return c


def _dft(rows: np.ndarray, n: int, sign: float) -> np.ndarray:
    """exp(sign * 2*pi*i * rows[:,None] * arange(n) / n) with the phase index
    reduced modulo n in integers (exact for awkward/prime n)."""
    idx = (rows[:, None].astype(np.int64) * np.arange(n, dtype=np.int64)[None, :]) % n
    return np.exp(sign * 2j * np.pi * idx / n)


def check_modes_2d(H, W, Ho, Wo, m1, m2):
    """Preconditions under which the reference's slicing is shape-consistent
    (integral_operators.py:137-142 docstring + slice semantics)."""
    if not (1 <= m1 <= min(H, Ho)):
        raise RuntimeError(f"modes1={m1} incompatible with grid rows {H}->{Ho}")
    if not (1 <= m2 <= min(W // 2 + 1, Wo // 2 + 1)):
        raise RuntimeError(f"modes2={m2} incompatible with grid cols {W}->{Wo}")


# --------------------------------------------------------------------------- #
# 2-D dense closed form (float64)
# --------------------------------------------------------------------------- #
def truncated_rfft2_dense(x: np.ndarray, m1: int, m2: int) -> np.ndarray:
    """X[b,i,j,l] = 1/(H W) sum_{h,w} x e^{-2 pi i (K_in[j] h/H + l w/W)}
    (integral_operators.py:187 restricted to the rows/cols :198-203 read)."""
    H, W = x.shape[-2:]
    Fh = _dft(corner_rows(H, m1), H, -1.0)          # (2m1, H)
    Fw = _dft(np.arange(m2), W, -1.0)               # (m2, W)
    x = x.astype(np.float64)
    t = np.einsum("bihw,lw->bihl", x, Fw)
    return np.einsum("jh,bihl->bijl", Fh, t) / (H * W)


def mix_modes_dense(X: np.ndarray, weights: list[np.ndarray]) -> np.ndarray:
    """einsum 'bixy,ioxy->boxy' per corner (integral_operators.py:178-179),
    corners concatenated along the row axis in weight order."""
    m1 = weights[0].shape[2]
    outs = []
    for c, w in enumerate(weights):
        outs.append(np.einsum("bijl,iojl->bojl", X[:, :, c * m1:(c + 1) * m1], w.astype(np.complex128)))
    return np.concatenate(outs, axis=2)


def truncated_irfft2_dense(O: np.ndarray, Ho: int, Wo: int, m1: int, m2: int) -> np.ndarray:
    """y = Re sum_{j,l} c_l keep_j O[j,l] e^{+2 pi i (K_out[j] h/Ho + l w/Wo)}
    (integral_operators.py:190-206: zero out_ft, ordered corner writes, irfft2
    with norm='forward' == unscaled inverse).  Real part is taken after the
    row-axis inverse, as a c2c-then-c2r inverse does."""
    Gh = _dft(corner_rows(Ho, m1), Ho, +1.0)        # (2m1, Ho)
    Gw = _dft(np.arange(m2), Wo, +1.0)              # (m2, Wo)
    keep = later_wins_mask(Ho, m1)
    c = hermitian_weights(Wo, m2)
    U = np.einsum("bojl,jh->bohl", O * keep[None, None, :, None], Gh)
    return np.einsum("bohl,lw->bohw", U * c[None, None, None, :], Gw).real


def spectral_conv2d_dense(x, w1, w2, Ho, Wo):
    """Full forward, float64 in / float64 out.  Returns (y, Xtrunc)."""
    x = np.asarray(x)
    w1 = np.asarray(w1)
    w2 = np.asarray(w2)
    H, W = x.shape[-2:]
    m1, m2 = w1.shape[2:]
    check_modes_2d(H, W, Ho, Wo, m1, m2)
    X = truncated_rfft2_dense(x, m1, m2)
    O = mix_modes_dense(X, [w1, w2])
    return truncated_irfft2_dense(O, Ho, Wo, m1, m2), X


def spectral_conv2d_dense_bwd(gy, X, w1, w2, H, W):
    """Hand-derived backward (PyTorch complex-grad convention
    grad = dL/dRe + i dL/dIm).  Returns (gx, gw1, gw2, gO, gX)."""
    gy = np.asarray(gy, dtype=np.float64)
    Ho, Wo = gy.shape[-2:]
    m1, m2 = w1.shape[2:]
    keep = later_wins_mask(Ho, m1)
    c = hermitian_weights(Wo, m2)
    Fh = _dft(corner_rows(Ho, m1), Ho, -1.0)
    Fw = _dft(np.arange(m2), Wo, -1.0)
    t = np.einsum("bohw,lw->bohl", gy, Fw)
    gO = np.einsum("jh,bohl->bojl", Fh, t) * c[None, None, None, :] * keep[None, None, :, None]
    Wt = np.concatenate([w1, w2], axis=2).astype(np.complex128)
    gWt = np.einsum("bijl,bojl->iojl", np.conj(X), gO)
    gX = np.einsum("iojl,bojl->bijl", np.conj(Wt), gO)
    Gh = _dft(corner_rows(H, m1), H, +1.0)
    Gw = _dft(np.arange(m2), W, +1.0)
    U = np.einsum("bijl,jh->bihl", gX, Gh)
    gx = np.einsum("bihl,lw->bihw", U, Gw).real / (H * W)
    return gx, gWt[:, :, :m1], gWt[:, :, m1:], gO, gX


# --------------------------------------------------------------------------- #
# 3-D dense closed form (float64)
# --------------------------------------------------------------------------- #
def check_modes_3d(H, W, T, Ho, Wo, To, m1, m2, m3):
    if not (1 <= m1 <= min(H, Ho)):
        raise RuntimeError(f"modes1={m1} incompatible with {H}->{Ho}")
    if not (1 <= m2 <= min(W, Wo)):
        raise RuntimeError(f"modes2={m2} incompatible with {W}->{Wo}")
    if not (1 <= m3 <= min(T // 2 + 1, To // 2 + 1)):
        raise RuntimeError(f"modes3={m3} incompatible with {T}->{To}")


def _weights3d_cat(ws, m1, m2):
    """(Ci,Co,2m1,2m2,m3) with weights1..4 placed at (lo,lo),(hi,lo),(lo,hi),(hi,hi)
    (integral_operators.py:410-421)."""
    w1, w2, w3, w4 = [np.asarray(w).astype(np.complex128) for w in ws]
    top = np.concatenate([w1, w3], axis=3)      # rows lo: cols lo | hi
    bot = np.concatenate([w2, w4], axis=3)      # rows hi
    return np.concatenate([top, bot], axis=2)


def _keep3d(Ho, Wo, m1, m2):
    """Later-wins mask over the (2m1, 2m2) corner grid for the assignment order
    w1 (lo,lo), w2 (hi,lo), w3 (lo,hi), w4 (hi,hi)."""
    rows = corner_rows(Ho, m1)
    cols = corner_rows(Wo, m2)
    owner = -np.ones((Ho, Wo), dtype=np.int64)
    order = [(0, 0), (1, 0), (0, 1), (1, 1)]    # (row corner, col corner)
    for k, (rc, cc) in enumerate(order):
        r = rows[rc * m1:(rc + 1) * m1]
        c = cols[cc * m2:(cc + 1) * m2]
        owner[np.ix_(r, c)] = k
    keep = np.zeros((2 * m1, 2 * m2))
    for k, (rc, cc) in enumerate(order):
        r = rows[rc * m1:(rc + 1) * m1]
        c = cols[cc * m2:(cc + 1) * m2]
        keep[rc * m1:(rc + 1) * m1, cc * m2:(cc + 1) * m2] = (owner[np.ix_(r, c)] == k)
    return keep


def truncated_rfft3_dense(x, m1, m2, m3):
    H, W, T = x.shape[-3:]
    Fh = _dft(corner_rows(H, m1), H, -1.0)
    Fw = _dft(corner_rows(W, m2), W, -1.0)
    Ft = _dft(np.arange(m3), T, -1.0)
    x = x.astype(np.float64)
    a = np.einsum("bihwt,nt->bihwn", x, Ft)
    a = np.einsum("kw,bihwn->bihkn", Fw, a)
    return np.einsum("jh,bihkn->bijkn", Fh, a) / (H * W * T)


def spectral_conv3d_dense(x, ws, Ho, Wo, To):
    """integral_operators.py:385-427 in closed form.  Returns (y, Xtrunc)."""
    x = np.asarray(x)
    H, W, T = x.shape[-3:]
    m1, m2, m3 = ws[0].shape[2:]
    check_modes_3d(H, W, T, Ho, Wo, To, m1, m2, m3)
    X = truncated_rfft3_dense(x, m1, m2, m3)
    Wt = _weights3d_cat(ws, m1, m2)
    O = np.einsum("bijkn,iojkn->bojkn", X, Wt)
    O = O * _keep3d(Ho, Wo, m1, m2)[None, None, :, :, None]
    c = hermitian_weights(To, m3)
    Gh = _dft(corner_rows(Ho, m1), Ho, +1.0)
    Gw = _dft(corner_rows(Wo, m2), Wo, +1.0)
    Gt = _dft(np.arange(m3), To, +1.0)
    U = np.einsum("bojkn,jh->bohkn", O, Gh)
    U = np.einsum("bohkn,kw->bohwn", U, Gw)
    y = np.einsum("bohwn,nt->bohwt", U * c, Gt).real
    return y, X


def spectral_conv3d_dense_bwd(gy, X, ws, H, W, T):
    gy = np.asarray(gy, dtype=np.float64)
    Ho, Wo, To = gy.shape[-3:]
    m1, m2, m3 = ws[0].shape[2:]
    c = hermitian_weights(To, m3)
    keep = _keep3d(Ho, Wo, m1, m2)
    Fh = _dft(corner_rows(Ho, m1), Ho, -1.0)
    Fw = _dft(corner_rows(Wo, m2), Wo, -1.0)
    Ft = _dft(np.arange(m3), To, -1.0)
    a = np.einsum("bohwt,nt->bohwn", gy, Ft)
    a = np.einsum("kw,bohwn->bohkn", Fw, a)
    gO = np.einsum("jh,bohkn->bojkn", Fh, a) * c * keep[None, None, :, :, None]
    Wt = _weights3d_cat(ws, m1, m2)
    gWt = np.einsum("bijkn,bojkn->iojkn", np.conj(X), gO)
    gX = np.einsum("iojkn,bojkn->bijkn", np.conj(Wt), gO)
    Gh = _dft(corner_rows(H, m1), H, +1.0)
    Gw = _dft(corner_rows(W, m2), W, +1.0)
    Gt = _dft(np.arange(m3), T, +1.0)
    U = np.einsum("bijkn,jh->bihkn", gX, Gh)
    U = np.einsum("bihkn,kw->bihwn", U, Gw)
    gx = np.einsum("bihwn,nt->bihwt", U, Gt).real / (H * W * T)
    gws = [gWt[:, :, :m1, :m2], gWt[:, :, m1:, :m2], gWt[:, :, :m1, m2:], gWt[:, :, m1:, m2:]]
    return gx, gws, gO, gX


# --------------------------------------------------------------------------- #
# FFT-sequence restatements (torch, differentiable) - the timed CPU baseline
# --------------------------------------------------------------------------- #
def spectral_conv2d_fft(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """rfft2(norm=forward) -> two corner contractions -> zero-padded irfft2
    (integral_operators.py:187-206).  Output is float32 (out_ft is cfloat)."""
    m1, m2 = w1.shape[2:]
    spec = torch.fft.rfft2(x, norm="forward")
    out = torch.zeros(x.shape[0], w1.shape[1], Ho, Wo // 2 + 1, dtype=torch.cfloat, device=x.device)
    for rows_in, rows_out, w in ((slice(None, m1), slice(None, m1), w1), (slice(-m1, None), slice(-m1, None), w2)):
        out[:, :, rows_out, :m2] = torch.einsum("bixy,ioxy->boxy", spec[:, :, rows_in, :m2], w)
    return torch.fft.irfft2(out, s=(Ho, Wo), norm="forward")


def spectral_conv3d_fft(x: torch.Tensor, ws, Ho: int, Wo: int, To: int) -> torch.Tensor:
    """integral_operators.py:398-426."""
    m1, m2, m3 = ws[0].shape[2:]
    spec = torch.fft.rfftn(x, dim=[-3, -2, -1], norm="forward")
    out = torch.zeros(x.shape[0], ws[0].shape[1], Ho, Wo, To // 2 + 1, dtype=torch.cfloat, device=x.device)
    lo1, hi1 = slice(None, m1), slice(-m1, None)
    lo2, hi2 = slice(None, m2), slice(-m2, None)
    for (r, c), w in zip(((lo1, lo2), (hi1, lo2), (lo1, hi2), (hi1, hi2)), ws):
        out[:, :, r, c, :m3] = torch.einsum("bixyz,ioxyz->boxyz", spec[:, :, r, c, :m3], w)
    return torch.fft.irfftn(out, s=(Ho, Wo, To), norm="forward")


def pointwise2d(x, weight, bias, Ho, Wo):
    """1x1 conv then bicubic anti-aliased resize (integral_operators.py:224-243)."""
    y = F.conv2d(x, weight, bias)
    return F.interpolate(y, size=(Ho, Wo), mode="bicubic", align_corners=True, antialias=True)


def pointwise3d(x, weight, bias, Ho, Wo, To):
    """1x1x1 conv, unnormalised rfftn, corner copy into an INPUT-sized zero
    spectrum, irfftn(s=out dims) and an identity trilinear resize
    (integral_operators.py:440-468, quirks kept)."""
    y = F.conv3d(x, weight, bias)
    ft = torch.fft.rfftn(y, dim=[-3, -2, -1])
    cut = torch.zeros_like(ft)
    a, b, c = Ho // 2, Wo // 2, To // 2
    for r in (slice(None, a), slice(-a, None)):
        for s in (slice(None, b), slice(-b, None)):
            cut[:, :, r, s, :c] = ft[:, :, r, s, :c]
    y = torch.fft.irfftn(cut, s=(Ho, Wo, To))
    return F.interpolate(y, size=(Ho, Wo, To), mode="trilinear", align_corners=True)


# --------------------------------------------------------------------------- #
# nn.Module forms (same parameter names/shapes as the reference so a reference
# state_dict loads strict=True); used for golden checks and the CPU baseline.
# --------------------------------------------------------------------------- #
class OracleSpectralConv2d(nn.Module):
    def __init__(self, in_codim, out_codim, dim1, dim2, modes1=None, modes2=None):
        super().__init__()
        ci, co = int(in_codim), int(out_codim)
        self.in_channels, self.out_channels = ci, co
        self.dim1, self.dim2 = dim1, dim2
        if modes1 is None:
            modes1, modes2 = dim1 // 2 - 1, dim2 // 2
        self.modes1, self.modes2 = modes1, modes2
        self.scale = (1 / (2 * ci)) ** 0.5
        self.weights1 = nn.Parameter(self.scale * torch.randn(ci, co, modes1, modes2, dtype=torch.cfloat))
        self.weights2 = nn.Parameter(self.scale * torch.randn(ci, co, modes1, modes2, dtype=torch.cfloat))

    def forward(self, x, dim1=None, dim2=None):
        if dim1 is not None:
            self.dim1, self.dim2 = dim1, dim2
        return spectral_conv2d_fft(x, self.weights1, self.weights2, self.dim1, self.dim2)


class OraclePointwise2d(nn.Module):
    def __init__(self, in_codim, out_codim, dim1, dim2):
        super().__init__()
        self.conv = nn.Conv2d(int(in_codim), int(out_codim), 1)
        self.dim1, self.dim2 = int(dim1), int(dim2)

    def forward(self, x, dim1=None, dim2=None):
        if dim1 is None:
            dim1, dim2 = self.dim1, self.dim2
        return pointwise2d(x, self.conv.weight, self.conv.bias, dim1, dim2)


class OracleOperatorBlock2d(nn.Module):
    def __init__(self, in_codim, out_codim, dim1, dim2, modes1, modes2, Normalize=False, Non_Lin=True):
        super().__init__()
        self.conv = OracleSpectralConv2d(in_codim, out_codim, dim1, dim2, modes1, modes2)
        self.w = OraclePointwise2d(in_codim, out_codim, dim1, dim2)
        self.normalize, self.non_lin = Normalize, Non_Lin
        if Normalize:
            self.normalize_layer = nn.InstanceNorm2d(int(out_codim), affine=True)

    def forward(self, x, dim1=None, dim2=None):
        y = self.conv(x, dim1, dim2) + self.w(x, dim1, dim2)
        if self.normalize:
            y = self.normalize_layer(y)
        return F.gelu(y) if self.non_lin else y


class OracleSpectralConv3d(nn.Module):
    def __init__(self, in_codim, out_codim, dim1, dim2, dim3, modes1=None, modes2=None, modes3=None):
        super().__init__()
        ci, co = int(in_codim), int(out_codim)
        self.in_channels, self.out_channels = ci, co
        self.dim1, self.dim2, self.dim3 = dim1, dim2, dim3
        if modes1 is None:
            modes1, modes2, modes3 = dim1, dim2, dim3 // 2 + 1
        self.modes1, self.modes2, self.modes3 = modes1, modes2, modes3
        self.scale = (1 / (2 * ci)) ** 0.5
        for k in range(1, 5):
            setattr(self, f"weights{k}", nn.Parameter(
                self.scale * torch.randn(ci, co, modes1, modes2, modes3, dtype=torch.cfloat)))

    def forward(self, x, dim1=None, dim2=None, dim3=None):
        if dim1 is not None:
            self.dim1, self.dim2, self.dim3 = dim1, dim2, dim3
        ws = [self.weights1, self.weights2, self.weights3, self.weights4]
        return spectral_conv3d_fft(x, ws, self.dim1, self.dim2, self.dim3)


class OraclePointwise3d(nn.Module):
    def __init__(self, in_codim, out_codim, dim1, dim2, dim3):
        super().__init__()
        self.conv = nn.Conv3d(int(in_codim), int(out_codim), 1)
        self.dim1, self.dim2, self.dim3 = int(dim1), int(dim2), int(dim3)

    def forward(self, x, dim1=None, dim2=None, dim3=None):
        if dim1 is None:
            dim1, dim2, dim3 = self.dim1, self.dim2, self.dim3
        return pointwise3d(x, self.conv.weight, self.conv.bias, dim1, dim2, dim3)


class OracleOperatorBlock3d(nn.Module):
    def __init__(self, in_codim, out_codim, dim1, dim2, dim3, modes1, modes2, modes3,
                 Normalize=False, Non_Lin=True):
        super().__init__()
        self.conv = OracleSpectralConv3d(in_codim, out_codim, dim1, dim2, dim3, modes1, modes2, modes3)
        self.w = OraclePointwise3d(in_codim, out_codim, dim1, dim2, dim3)
        self.normalize, self.non_lin = Normalize, Non_Lin
        if Normalize:
            self.normalize_layer = nn.InstanceNorm3d(int(out_codim), affine=True)

    def forward(self, x, dim1=None, dim2=None, dim3=None):
        y = self.conv(x, dim1, dim2, dim3) + self.w(x, dim1, dim2, dim3)
        if self.normalize:
            y = self.normalize_layer(y)
        return F.gelu(y) if self.non_lin else y


# --------------------------------------------------------------------------- #
# harness pieces of the reference training step (loss + optimiser)
# --------------------------------------------------------------------------- #
def lp_loss_rel_sum(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """LpLoss(size_average=False).rel (utilities3.py:86-100): sum over the batch of
    ||pred-target||_2 / ||target||_2."""
    n = pred.shape[0]
    d = torch.norm(pred.reshape(n, -1) - target.reshape(n, -1), 2, 1)
    return torch.sum(d / torch.norm(target.reshape(n, -1), 2, 1))


def reference_adam_step(params, grads, exp_avgs, exp_avg_sqs, step, lr, beta1, beta2, eps, weight_decay):
    """One step of the reference's Adam (Adam.py:27-52): coupled L2 decay, second
    moment from g*conj(g) (complex modulus), bias-corrected."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        if weight_decay != 0:
            g = g + weight_decay * p
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g.conj(), value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)
