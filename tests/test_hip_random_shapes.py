"""Seeded random geometries through the C ABI against float64 references - the shapes no hand-written list thinks of (the first run
of the K2 version of this test found an odd-channel-count bug in a kernel variant that every listed shape missed).  pytest -m gpu"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
TOL = 2e-5


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _spectral2d_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        H, W = int(rng.integers(4, 90)), int(rng.integers(4, 90))
        Ho, Wo = int(rng.integers(4, 90)), int(rng.integers(4, 90))
        m1 = int(rng.integers(1, min(H, Ho) + 1))                      # up to full rows: overlapping corners included
        m2 = int(rng.integers(1, min(W, Wo) // 2 + 2))
        B, Ci, Co = int(rng.integers(1, 4)), int(rng.integers(1, 7)), int(rng.integers(1, 7))
        try:
            so.check_modes_2d(H, W, Ho, Wo, m1, m2)
        except Exception:
            continue
        out.append((B, Ci, Co, H, W, Ho, Wo, m1, m2))
    return out


@pytest.mark.parametrize("cfg", _spectral2d_cases(24, 4242), ids=lambda c: "-".join(map(str, c)))
def test_spectral_conv2d_random_geometry(cfg):
    """SpectralConv2d forward / backward (K1, K2, K3 in whichever form the shape selects) against the dense float64 oracle."""
    from uno_amd.integral_operators import spectral_conv2d
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = cfg
    rng = np.random.default_rng(sum(cfg))
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    sc = (1 / (2 * Ci)) ** 0.5
    w1, w2 = [(sc * (rng.standard_normal((Ci, Co, m1, m2)) + 1j * rng.standard_normal((Ci, Co, m1, m2)))).astype(np.complex64) for _ in range(2)]
    gy = rng.standard_normal((B, Co, Ho, Wo)).astype(np.float32)
    y_ref, X = so.spectral_conv2d_dense(x, w1, w2, Ho, Wo)
    gx_ref, gw1_ref, gw2_ref = so.spectral_conv2d_dense_bwd(gy, X, w1, w2, H, W)[:3]
    xd, w1d, w2d = cu(x).requires_grad_(True), cu(w1).requires_grad_(True), cu(w2).requires_grad_(True)
    y = spectral_conv2d(xd, w1d, w2d, Ho, Wo)
    y.backward(cu(gy))
    assert rel_err(y.detach().cpu().numpy(), y_ref) < TOL
    assert rel_err(xd.grad.cpu().numpy(), gx_ref) < TOL
    assert rel_err(w1d.grad.cpu().numpy(), gw1_ref) < TOL
    assert rel_err(w2d.grad.cpu().numpy(), gw2_ref) < TOL


def _channel_cases(n, seed):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 5)), int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.choice([1, 3, 17, 128, 777, 1024, 1029, 4100, 9000])))
            for _ in range(n)]


@pytest.mark.parametrize("cfg", _channel_cases(24, 99), ids=lambda c: "-".join(map(str, c)))
def test_channel_mix_random_shapes(cfg):
    """K8 (forward, transposed, accumulating) and K9 on random (batch, channels, pixels) against float64 matmuls."""
    from uno_amd import _native
    B, Ci, Co, P = cfg
    g = torch.Generator().manual_seed(B + 7 * Ci + 13 * Co + P)
    x = torch.randn(B, Ci, P, generator=g)
    w = torch.randn(Co, Ci, generator=g) / Ci ** 0.5
    b = torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, P, generator=g)
    xd, wd, bd, gyd = x.cuda(), w.cuda(), b.cuda(), gy.cuda()
    y_ref = torch.matmul(w.double(), x.double()) + b.double().view(1, -1, 1)
    assert rel_err(_native.channel_mix(xd, wd, bd).cpu().numpy(), y_ref.numpy()) < 1e-5
    gx_ref = torch.matmul(w.double().t(), gy.double())
    assert rel_err(_native.channel_mix(gyd, wd, None, transpose_w=True).cpu().numpy(), gx_ref.numpy()) < 1e-5
    out = y_ref.float().cuda().clone()
    _native.channel_mix(xd, wd, None, out=out)
    assert rel_err(out.cpu().numpy(), (2 * y_ref - b.double().view(1, -1, 1)).numpy()) < 1e-5
    gw, gb = _native.channel_wgrad(gyd, xd)
    assert rel_err(gw.cpu().numpy(), torch.einsum("bop,bip->oi", gy.double(), x.double()).numpy()) < 1e-5
    assert rel_err(gb.cpu().numpy(), gy.double().sum(dim=(0, 2)).numpy()) < 1e-5


def _resample_cases(n, seed):
    rng = np.random.default_rng(seed)
    return [((int(rng.integers(2, 100)), int(rng.integers(2, 100))), (int(rng.integers(2, 100)), int(rng.integers(2, 100))), int(rng.integers(1, 20)))
            for _ in range(n)]


@pytest.mark.parametrize("cfg", _resample_cases(20, 31), ids=lambda c: f"{c[0][0]}x{c[0][1]}-{c[1][0]}x{c[1][1]}-n{c[2]}")
def test_resample_random_sizes(cfg):
    """K7 (both forms of its row pass, with and without accumulation) against torch's CPU bicubic anti-aliased interpolation."""
    from uno_amd.resample import resample2d_bicubic_aa, resample_forward
    (H, W), (Ho, Wo), n = cfg
    g = torch.Generator().manual_seed(H * 31 + Wo + n)
    x = torch.randn(n, 1, H, W, generator=g)
    gy = torch.randn(n, 1, Ho, Wo, generator=g)
    xc = x.clone().requires_grad_(True)
    yc = F.interpolate(xc, size=(Ho, Wo), mode="bicubic", align_corners=True, antialias=True)
    yc.backward(gy)
    xd = x.cuda().requires_grad_(True)
    yd = resample2d_bicubic_aa(xd, Ho, Wo)
    yd.backward(gy.cuda())
    assert rel_err(yd.detach().cpu().numpy(), yc.detach().numpy()) < 5e-6
    assert rel_err(xd.grad.cpu().numpy(), xc.grad.numpy()) < 5e-6
    acc = gy.cuda().clone()
    resample_forward(x.cuda(), Ho, Wo, out=acc)
    assert rel_err(acc.cpu().numpy(), (gy + yc.detach()).numpy()) < 5e-6


def _rel64(a, b):
    d = (a.double().cpu() - b.cpu()).norm().item()
    n = b.norm().item()
    return d / n if n > 0 else d


def _fused_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nd = int(rng.choice([1, 2, 2, 3]))
        sp = tuple(int(v) for v in rng.integers(2, [4000, 120, 24][nd - 1] + 1, size=nd))
        out.append((int(rng.integers(1, 4)), int(rng.integers(1, 70)), sp))
    return out


@pytest.mark.parametrize("cfg", _fused_cases(16, 5), ids=lambda c: f"B{c[0]}-C{c[1]}-" + "x".join(map(str, c[2])))
def test_fused_pointwise_kernels_random_shapes(cfg):
    """K13 InstanceNorm(affine) + GELU, K11 GELU + projection, K12 GELU + padding (2-D shapes) on random shapes, forward and every
    gradient against the stock op sequences in float64."""
    import torch.nn as nn
    from uno_amd.integral_operators import gelu_pad2d, gelu_project, instance_norm_gelu
    B, C, sp = cfg
    g = torch.Generator().manual_seed(B + 3 * C + sum(sp))
    x = 2.0 * torch.randn(B, C, *sp, generator=g) + 0.5
    gy = torch.randn(B, C, *sp, generator=g)
    # K13
    cls = {1: nn.InstanceNorm1d, 2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}[len(sp)]
    norm = cls(C, affine=True)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g)); norm.bias.copy_(torch.randn(C, generator=g))
    ref_norm = cls(C, affine=True).double()
    ref_norm.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    x2 = x.double().requires_grad_(True)
    y2 = F.gelu(ref_norm(x2))
    ref = torch.autograd.grad(y2, [x2] + list(ref_norm.parameters()), gy.double())
    norm = norm.cuda()
    xd = x.cuda().requires_grad_(True)
    y = instance_norm_gelu(xd, norm, True)
    got = torch.autograd.grad(y, [xd] + list(norm.parameters()), gy.cuda())
    assert _rel64(y, y2.detach()) < 5e-6
    for a, r in zip(got, ref):
        assert a.shape == r.shape and _rel64(a, r) < 3e-5
    # K11
    w = torch.randn(1, C, generator=g); b = torch.randn(1, generator=g)
    pre = x.cuda().requires_grad_(True); wd = w.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True)
    yp = gelu_project(pre, wd, bd)
    gyp = torch.randn(B, 1, *sp, generator=g)
    gotp = torch.autograd.grad(yp, (pre, wd, bd), gyp.cuda())
    pre2, w2, b2 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yp2 = torch.einsum("oc,bc...->bo...", w2, F.gelu(pre2)) + b2.view(1, 1, *([1] * len(sp)))
    refp = torch.autograd.grad(yp2, (pre2, w2, b2), gyp.double())
    assert _rel64(yp, yp2.detach()) < 5e-6
    for a, r in zip(gotp, refp):
        assert a.shape == r.shape and _rel64(a, r) < 3e-5
    # K12
    if len(sp) == 2:
        pad = (int(sp[0] % 7), int(sp[1] % 5))
        s = x.cuda().requires_grad_(True)
        yq = gelu_pad2d(s, pad[0], pad[1])
        gyq = torch.randn(B, C, sp[0] + pad[0], sp[1] + pad[1], generator=g)
        (gs,) = torch.autograd.grad(yq, s, gyq.cuda())
        s2 = x.double().requires_grad_(True)
        yq2 = F.pad(F.gelu(s2), [0, pad[1], 0, pad[0]])
        (gs2,) = torch.autograd.grad(yq2, s2, gyq.double())
        assert _rel64(yq, yq2.detach()) < 5e-6 and _rel64(gs, gs2) < 5e-6


def _plane3d_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        din = (int(rng.integers(2, 30)), int(rng.integers(2, 30)), int(rng.integers(2, 24)))
        dout = (int(rng.integers(2, 30)), int(rng.integers(2, 30)), int(rng.integers(2, 24)))
        m1 = int(rng.integers(1, min(din[0], dout[0]) + 1))
        m2 = int(rng.integers(1, min(din[1], dout[1]) + 1))
        m3 = int(rng.integers(1, min(din[2], dout[2]) // 2 + 2))
        try:
            so.check_modes_3d(*din, *dout, m1, m2, m3)
        except Exception:
            continue
        out.append((int(rng.integers(1, 3)), int(rng.integers(1, 5)), int(rng.integers(1, 5)), din, dout, (m1, m2, m3)))
    return out


@pytest.mark.parametrize("cfg", _plane3d_cases(14, 808), ids=lambda c: f"B{c[0]}-{c[1]}x{c[2]}-" + "x".join(map(str, c[3])) + "-" + "x".join(map(str, c[4])) + "-m" + "x".join(map(str, c[5])))
def test_spectral_conv3d_random_geometry_small_batches(cfg):
    """SpectralConv3d with few volumes (the plane-batched K1p / K3p + K5 / K6 path, the any-mode forms, overlapping corners):
    forward and all gradients against the dense float64 oracle."""
    from uno_amd.spectral3d import spectral_conv3d
    B, Ci, Co, din, dout, modes = cfg
    rng = np.random.default_rng(sum(din) * 7 + sum(dout) + sum(modes))
    x = rng.standard_normal((B, Ci, *din)).astype(np.float32)
    sc = (1 / (2 * Ci)) ** 0.5
    ws = [(sc * (rng.standard_normal((Ci, Co, *modes)) + 1j * rng.standard_normal((Ci, Co, *modes)))).astype(np.complex64) for _ in range(4)]
    gy = rng.standard_normal((B, Co, *dout)).astype(np.float32)
    y_ref, X = so.spectral_conv3d_dense(x, ws, *dout)
    gx_ref, gws_ref, _, _ = so.spectral_conv3d_dense_bwd(gy, X, ws, *din)
    xd = cu(x).requires_grad_(True)
    wd = [cu(w).requires_grad_(True) for w in ws]
    y = spectral_conv3d(xd, wd, *dout)
    y.backward(cu(gy))
    assert rel_err(y.detach().cpu().numpy(), y_ref) < TOL
    assert rel_err(xd.grad.cpu().numpy(), gx_ref) < TOL
    for k in range(4):
        assert rel_err(wd[k].grad.cpu().numpy(), gws_ref[k]) < TOL, k
