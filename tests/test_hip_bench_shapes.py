"""Every spectral kernel that a bench table names is reached by an oracle comparison AT THE BENCH SHAPE.  pytest -m gpu

The kernels of the spectral path are picked per shape (register / full-tile / half-tile transforms, plane-batched or one workgroup
per volume, 4x4x1 or LDS-staged per-mode GEMM, 8- or 16-mode variants).  The shape-class tests elsewhere cover every variant, but a
variant can still be wrong only at the geometry a benchmark dispatches it with.  Here the layers of the four workloads bench.py
times (BASELINE.json configs[1..4]) run at their full sizes - batch, channels, grid, modes - against the reference's op sequence
(rfft -> corner einsum -> irfft, oracle/spectral_oracle.py, pinned by the goldens) on the host, with the library's per-kernel
records on: the names that ran are collected, and the last test asserts that every spectral kernel named in the committed bench
line (profiles/rNN_bench_n1.json) is among them.  Tolerance: relative L2 <= 2e-5 (float32 spectral path), 5e-5 for whole blocks.

Reference: integral_operators.py:181-207 (2-D), :385-427 (3-D); callers darcy_flow_uno2d.py:108-125, navier_stokes_uno2d.py:160-214,
navier_stokes_uno3d.py:239-409."""
import glob
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
TOL = 2e-5
COVERED = set()          # spectral kernel names that ran inside an oracle-compared call of this module


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _spectral(name):
    return "dft" in name or "mode_gemm" in name


def _run_profiled(fn):
    from uno_amd import _native
    _native.profile_begin(4096)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        ran = [n for n, _, _ in _native.profile_end()]
    return out, ran


def _check2d(B, Ci, Co, H, W, Ho, Wo, m1, m2, seed):
    from uno_amd.integral_operators import spectral_conv2d
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, W, generator=g)
    sc = (1 / (2 * Ci)) ** 0.5
    w1 = sc * torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g)
    w2 = sc * torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g)
    gy = torch.randn(B, Co, Ho, Wo, generator=g)
    xr, w1r, w2r = (t.clone().requires_grad_(True) for t in (x, w1, w2))
    so.spectral_conv2d_fft(xr, w1r, w2r, Ho, Wo).backward(gy)
    y_ref = so.spectral_conv2d_fft(x, w1, w2, Ho, Wo)
    xd, w1d, w2d = (t.to(dev()).requires_grad_(True) for t in (x, w1, w2))

    def run():
        y = spectral_conv2d(xd, w1d, w2d, Ho, Wo)
        y.backward(gy.to(dev()))
        return y
    y, ran = _run_profiled(run)
    assert rel_err(y.detach().cpu().numpy(), y_ref.numpy()) < TOL
    assert rel_err(xd.grad.cpu().numpy(), xr.grad.numpy()) < TOL
    assert rel_err(w1d.grad.cpu().numpy(), w1r.grad.numpy()) < TOL
    assert rel_err(w2d.grad.cpu().numpy(), w2r.grad.numpy()) < TOL
    names = {n for n in ran if _spectral(n)}
    assert names, ran
    COVERED.update(names)
    return names


def _check3d(B, Ci, Co, din, dout, modes, seed):
    from uno_amd.spectral3d import spectral_conv3d
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, *din, generator=g)
    sc = (1 / (2 * Ci)) ** 0.5
    ws = [sc * torch.randn(Ci, Co, *modes, dtype=torch.cfloat, generator=g) for _ in range(4)]
    gy = torch.randn(B, Co, *dout, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in ws]
    y_ref = so.spectral_conv3d_fft(xr, wr, *dout)
    y_ref.backward(gy)
    xd = x.to(dev()).requires_grad_(True)
    wd = [w.to(dev()).requires_grad_(True) for w in ws]

    def run():
        y = spectral_conv3d(xd, wd, *dout)
        y.backward(gy.to(dev()))
        return y
    y, ran = _run_profiled(run)
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < TOL
    assert rel_err(xd.grad.cpu().numpy(), xr.grad.numpy()) < TOL
    for k in range(4):
        assert rel_err(wd[k].grad.cpu().numpy(), wr[k].grad.numpy()) < TOL, k
    names = {n for n in ran if _spectral(n)}
    assert names, ran
    COVERED.update(names)
    return names, y, xd, gy


# ------------------------------------------------------------------ C2: Darcy 421^2, UNO_9(3,64,pad=5), batch 16 (padded grid 446)
D = 446
C2_LAYERS = {   # (Ci, Co, H -> Ho, modes) of darcy_flow_uno2d.py:108-121 at width 64
    "block":  (64, 64, 421, 421, 20),           # BASELINE's roofline block
    "conv0": (64, 128, D, D // 2, 18),
    "conv1": (128, 256, D // 2, D // 4, 8),
    "conv2": (256, 256, D // 4, D // 4, 8),
    "conv4": (256, 128, D // 4, D // 2, 8),
}


@pytest.mark.parametrize("layer", list(C2_LAYERS))
def test_c2_darcy_layers_full_size(layer):
    Ci, Co, H, Ho, m = C2_LAYERS[layer]
    names = _check2d(16, Ci, Co, H, H, Ho, Ho, m, m, seed=H + Ho + Ci)
    if layer == "block":       # the kernels the headline roofline figure is quoted on
        assert any("dft2d_fwd_ht_kernel" in n for n in names) and any("dft2d_inv_ft_kernel" in n for n in names), names


def test_c2_darcy_conv5_two_source_block_full_size():
    """conv5 = OperatorBlock_2D(256, 64, 446, 446, 18, 18) on cat([conv4 out, c0]) (darcy_flow_uno2d.py:117-121) through the
    two-source form the harness model uses (grouped K1, stage-level K2 / K3, two-source channel mix, up-sampling K7), with the
    GELU deferred: the pre-activation sum against the oracle block's two branches on the host, forward and every gradient."""
    from uno_amd.integral_operators import OperatorBlock_2D
    torch.manual_seed(5)
    B, C1, C2, Co, H, Ho, m = 16, 128, 128, 64, D // 2, D, 18
    ob = so.OracleOperatorBlock2d(C1 + C2, Co, Ho, Ho, m, m)
    blk = OperatorBlock_2D(C1 + C2, Co, Ho, Ho, m, m)
    blk.load_state_dict(ob.state_dict(), strict=True)
    blk = blk.to(dev())
    g = torch.Generator().manual_seed(55)
    x1, x2 = torch.randn(B, C1, H, H, generator=g), torch.randn(B, C2, H, H, generator=g)
    gy = torch.randn(B, Co, Ho, Ho, generator=g)
    xc = torch.cat([x1, x2], 1).requires_grad_(True)
    pre_ref = ob.conv(xc, Ho, Ho) + ob.w(xc, Ho, Ho)
    pre_ref.backward(gy)
    x1d, x2d = x1.to(dev()).requires_grad_(True), x2.to(dev()).requires_grad_(True)

    def run():
        pre = blk.forward_cat([x1d, x2d], Ho, Ho, defer_gelu=True)
        pre.backward(gy.to(dev()))
        return pre
    pre, ran = _run_profiled(run)
    assert rel_err(pre.detach().cpu().numpy(), pre_ref.detach().numpy()) < 5e-5
    gx = xc.grad.numpy()
    assert rel_err(x1d.grad.cpu().numpy(), gx[:, :C1]) < 5e-5
    assert rel_err(x2d.grad.cpu().numpy(), gx[:, C1:]) < 5e-5
    refp = dict(ob.named_parameters())
    for k, p in blk.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), refp[k].grad.numpy()) < 5e-5, k
    COVERED.update(n for n in ran if _spectral(n))




K3A_CASES = {   # (images, source grid, output grid, modes, adjoint operators, hermitian columns + overlap mask, scale) of the fused
                # "inverse transform + resampled point-wise branch" launches of ONE Darcy step (uno_dft2d_inverse_add)
    "conv4_forward": (16 * 128, 111, 223, 8, False, True, 1.0),
    "conv5_forward": (16 * 64, 223, 446, 18, False, True, 1.0),
    "conv0_input_gradient": (16 * 64, 223, 446, 18, True, False, 1.0 / (446 * 446)),
    # conv1's input gradient carries conv5's deferred c0 spectrum (GradJoin.merge): modes 18 on the 223^2 grid
    "conv1_input_gradient_merged_spectrum": (16 * 128, 111, 223, 18, True, False, 1.0 / (223 * 223)),
    "conv1_input_gradient": (16 * 128, 111, 223, 8, True, False, 1.0 / (223 * 223)),
}


@pytest.mark.parametrize("case", list(K3A_CASES))
def test_c2_darcy_fused_inverse_add_full_size(case):
    """K3-A at the geometries the headline step launches it with (reference integral_operators.py:272-273 + :240-242: inverse transform
    of the corner spectrum + bicubic / align_corners / antialias resampling of the low-resolution point-wise result, and the adjoint
    for the input gradient of a down-sampling block), the whole launch at bench size, 24 images spread over it against float64."""
    from uno_amd import _native
    from uno_amd import resample as rs
    n, Hs, H, m, adjoint, herm, scale = K3A_CASES[case]
    d = dev()
    g = torch.Generator().manual_seed(n + H + m)
    spec = torch.randn(n, 2 * m, m, dtype=torch.complex64, generator=g)
    t = torch.randn(n, Hs, Hs, generator=g)
    tabs = rs.upsample_add_tables(Hs, Hs, H, H, str(d), adjoint)
    assert tabs is not None and _native.dft2d_inverse_add_applies(n, H, H, m, m, Hs, Hs)
    y, ran = _run_profiled(lambda: _native.dft2d_inverse(spec.to(d), H, H, scale, herm, herm, addend=(t.to(d), tabs)))
    names = {k for k in ran if _spectral(k)}
    assert any("dft2d_inv_ft_add_kernel" in k for k in names), ran
    # float64 reference of the sampled images: dense inverse DFT over the two corners + the dense resampling operators
    R = (rs._matrix(H, Hs).t() if adjoint else rs._matrix(Hs, H)).double().numpy()          # (H, Hs), square grids: rows == columns
    K = np.array([j if j < m else H - 2 * m + j for j in range(2 * m)])
    Eh = np.exp(2j * np.pi * np.outer(np.arange(H), K) / H)                                   # (H, 2m)
    Ew = np.exp(2j * np.pi * np.outer(np.arange(m), np.arange(H)) / H)                        # (m, W)
    c = np.array([1.0 if (l == 0 or 2 * l == H) else 2.0 for l in range(m)]) if herm else np.ones(m)
    keep = np.array([0.0 if (herm and j < m and j >= H - m) else 1.0 for j in range(2 * m)])
    idx = sorted(set(np.linspace(0, n - 1, 24).astype(int).tolist()))
    yk = y[idx].cpu().double().numpy()
    for q, i in enumerate(idx):
        O = spec[i].numpy().astype(np.complex128) * scale * c[None, :] * keep[:, None]
        ref = (Eh @ O @ Ew).real + R @ t[i].double().numpy() @ R.T
        assert rel_err(yk[q], ref) < TOL, (case, i, rel_err(yk[q], ref))
    COVERED.update(names)

# ------------------------------------------------------------------ C3: NS-2D UNO(14,32), 64^2, batch 32 (navier_stokes_uno2d.py:160-214)
C3_LAYERS = [   # Ci, Co, H, Ho, modes  
    (32, 48, 64, 48, 22), (48, 96, 48, 32, 14), (96, 192, 32, 16, 6), (192, 192, 16, 16, 6),
    (192, 96, 16, 32, 6), (192, 48, 32, 48, 14), (96, 32, 48, 64, 22),
]


@pytest.mark.parametrize("cfg", C3_LAYERS, ids=lambda c: "x".join(map(str, c)))
def test_c3_ns2d_layers_full_width(cfg):
    Ci, Co, H, Ho, m = cfg
    _check2d(32, Ci, Co, H, H, Ho, Ho, m, m, seed=Ci + Co + H)


@pytest.mark.parametrize("cfg", C3_LAYERS, ids=lambda c: "x".join(map(str, c)))
def test_c3_ns2d_accumulating_weight_gradient(cfg):
    """The roll-out sums 40 weight gradients per layer in place (uno_mode_wgrad_acc, beta = 1: separate kernel instantiations):
    gw += sum_b conj(X[b, i]) gO[b, o] per mode at the full-width layer shapes, against float64 on the host."""
    from uno_amd import _native
    Ci, Co, H, Ho, m = cfg
    B = 32
    g = torch.Generator().manual_seed(Ci * 3 + Co + m)
    xt = torch.randn(B, Ci, 2 * m, m, dtype=torch.cfloat, generator=g)
    gO = torch.randn(B, Co, 2 * m, m, dtype=torch.cfloat, generator=g)
    base = [torch.randn(Ci, Co, m, m, dtype=torch.cfloat, generator=g) for _ in range(2)]
    ref = [base[c].to(torch.complex128) + torch.einsum("bixy,boxy->ioxy", xt[:, :, c * m:(c + 1) * m].conj().to(torch.complex128),
                                                         gO[:, :, c * m:(c + 1) * m].to(torch.complex128)) for c in range(2)]
    out = [b.clone().to(dev()) for b in base]

    def run():
        return _native.mode_wgrad(xt.to(dev()), gO.to(dev()), (Ci, Co, m, m), 2, out=out, accumulate=True)
    _, ran = _run_profiled(run)
    for c in range(2):
        assert rel_err(torch.view_as_real(out[c]).cpu().numpy(), torch.view_as_real(ref[c]).numpy()) < TOL, c
    names = {n for n in ran if _spectral(n)}
    assert names and all(n.rstrip(">").endswith("true") for n in names), names      # the accumulating instantiations
    COVERED.update(names)


@pytest.mark.parametrize("cfg", C3_LAYERS, ids=lambda c: "x".join(map(str, c)))
def test_c3_ns2d_weight_gradient_batched_over_the_rollout(cfg):
    """From the second training step on the roll-out's 40 weight gradients of a layer are ONE per-mode GEMM with K = 40 x 32 over
    the layer's stacked spectra (integral_operators.TIME_BATCHED_WGRAD): the kernels that call dispatches at the full-width layer
    shapes, against float64 on the host."""
    from uno_amd import _native
    Ci, Co, H, Ho, m = cfg
    T, B = 40, 32
    g = torch.Generator().manual_seed(Ci * 5 + Co + m)
    xt = torch.randn(T * B, Ci, 2 * m, m, dtype=torch.cfloat, generator=g)
    gO = torch.randn(T * B, Co, 2 * m, m, dtype=torch.cfloat, generator=g)
    xd, gd = xt.to(dev()), gO.to(dev())
    ref = [torch.einsum("bixy,boxy->ioxy", xd[:, :, c * m:(c + 1) * m].conj().to(torch.complex128),
                        gd[:, :, c * m:(c + 1) * m].to(torch.complex128)).cpu() for c in range(2)]     # f64 einsum on the device: 40x the host time otherwise
    refh = torch.einsum("bixy,boxy->ioxy", xt[:64, :, :m].conj().to(torch.complex128), gO[:64, :, :m].to(torch.complex128))
    chk = torch.einsum("bixy,boxy->ioxy", xd[:64, :, :m].conj().to(torch.complex128), gd[:64, :, :m].to(torch.complex128)).cpu()
    assert rel_err(torch.view_as_real(chk).numpy(), torch.view_as_real(refh).numpy()) < 1e-12      # the device f64 reference agrees with the host's
    out, ran = _run_profiled(lambda: _native.mode_wgrad(xd, gd, (Ci, Co, m, m), 2))
    for c in range(2):
        assert rel_err(torch.view_as_real(out[c]).cpu().numpy(), torch.view_as_real(ref[c]).numpy()) < TOL, c
    names = {n for n in ran if _spectral(n)}
    assert names, ran
    COVERED.update(names)


# ------------------------------------------------------------------ C4: NS-3D (block + Uno3D_T20 layers), batch 8
def test_c4_block_full_size_volume_kernels():
    """SpectralConv3d(32,32,64,64,20,16,16,8), batch 8 = 256 volumes: the one-workgroup-per-volume kernels bench.py's 3-D block
    figure is quoted on - y, gx, gw1..4 against the reference op sequence, adjoint identity and determinism."""
    from uno_amd.spectral3d import spectral_conv3d
    names, y, xd, gy = _check3d(8, 32, 32, (64, 64, 20), (64, 64, 20), (16, 16, 8), seed=4)
    assert any("dft3d_fwd_volume_kernel" in n for n in names) and any("dft3d_inv_volume_kernel" in n for n in names), names
    with torch.no_grad():
        a = torch.dot(y.detach().double().flatten(), gy.to(dev()).double().flatten())
        b = torch.dot(xd.detach().double().flatten(), xd.grad.double().flatten())
        assert abs(a.item() - b.item()) <= 1e-5 * max(abs(a.item()), abs(b.item()))


def _t20_layers(w):
    # (Ci, Co, din, dout, modes) of Uno3D_T20(6, w, pad=3) on (8, 64, 64, 10): navier_stokes_uno3d.py:263-285, 329-369
    return [
        (w, 2 * w, (64, 64, 13), (48, 48, 13), (22, 22, 5)), (2 * w, 4 * w, (48, 48, 13), (32, 32, 13), (14, 14, 5)),
        (4 * w, 8 * w, (32, 32, 13), (16, 16, 15), (6, 6, 5)), (8 * w, 16 * w, (16, 16, 15), (16, 16, 15), (6, 6, 6)),
        (16 * w, 4 * w, (16, 16, 15), (32, 32, 23), (6, 6, 6)), (8 * w, 2 * w, (32, 32, 23), (48, 48, 26), (14, 14, 8)),
        (4 * w, 2 * w, (48, 48, 26), (64, 64, 26), (22, 22, 8)),
    ]


@pytest.mark.parametrize("w", [8, 32])
@pytest.mark.parametrize("layer", range(7))
def test_c4_ns3d_model_layers_full_size(w, layer):
    Ci, Co, din, dout, modes = _t20_layers(w)[layer]
    _check3d(8, Ci, Co, din, dout, modes, seed=w + layer)


@pytest.mark.parametrize("w", [8, 32])
@pytest.mark.parametrize("layer", range(7))
def test_c4_ns3d_pointwise_resample_full_size(w, layer):
    """The FFT crop / resample of pointwise_op_3D (reference integral_operators.py:448-463) at the NS-3D model's own shapes - it runs
    on the same pruned-DFT kernel families (plane transforms + leading-axis cdft kernels with explicit frequency tables), which the
    bench's kernel lists therefore name: forward and input gradient against the reference's op sequence in float64 on the host."""
    from uno_amd.integral_operators import _FftResample3dFn, _resample3d_plan
    _, Co, din, dout, _ = _t20_layers(w)[layer]
    plan = _resample3d_plan(din, dout, dev())
    if plan is None:
        pytest.skip("outside the pruned-DFT resampling kernels' range: the model runs stock rocFFT here")
    g = torch.Generator().manual_seed(w * 10 + layer)
    x = torch.randn(8, Co, *din, generator=g)
    gy = torch.randn(8, Co, *dout, generator=g)
    xr = x.double().requires_grad_(True)
    spec = torch.fft.rfftn(xr, dim=[-3, -2, -1])
    kept = torch.zeros_like(spec)
    h1, h2, h3 = dout[0] // 2, dout[1] // 2, dout[2] // 2
    for rows in (slice(None, h1), slice(-h1, None)):
        for cols in (slice(None, h2), slice(-h2, None)):
            kept[:, :, rows, cols, :h3] = spec[:, :, rows, cols, :h3]
    yr = torch.fft.irfftn(kept, s=dout)
    yr.backward(gy.double())
    xd = x.to(dev()).requires_grad_(True)

    def run():
        y = _FftResample3dFn.apply(xd, dout, plan)
        y.backward(gy.to(dev()))
        return y
    y, ran = _run_profiled(run)
    assert rel_err(y.detach().cpu().numpy(), yr.detach().numpy()) < TOL
    assert rel_err(xd.grad.cpu().numpy(), xr.grad.numpy()) < TOL
    COVERED.update(n for n in ran if _spectral(n))


@pytest.mark.parametrize("w", [8, 32])
@pytest.mark.parametrize("layer", range(7))
def test_c4_ns3d_pointwise_resample_accumulating_full_size(w, layer):
    """The ACCUMULATING form of the same resampling (uno_fft_resample3d_acc: what the one-buffer OperatorBlock_3D runs - the point-wise
    branch's last transform adds into the spectral branch's output and writes the GELU in the same pass, reference
    integral_operators.py:506-512) at the NS-3D model's shapes: s + resample(t) and gelu(s + resample(t)) against float64 on the host."""
    from uno_amd import _native
    from uno_amd.integral_operators import _resample3d_plan
    _, Co, din, dout, _ = _t20_layers(w)[layer]
    plan = _resample3d_plan(din, dout, dev())
    if plan is None:
        pytest.skip("outside the pruned-DFT resampling kernels' range: the model runs stock rocFFT here")
    g = torch.Generator().manual_seed(w * 10 + layer + 500)
    t = torch.randn(8, Co, *din, generator=g)
    s0 = torch.randn(8, Co, *dout, generator=g)
    spec = torch.fft.rfftn(t.double(), dim=[-3, -2, -1])
    kept = torch.zeros_like(spec)
    h1, h2, h3 = dout[0] // 2, dout[1] // 2, dout[2] // 2
    for rows in (slice(None, h1), slice(-h1, None)):
        for cols in (slice(None, h2), slice(-h2, None)):
            kept[:, :, rows, cols, :h3] = spec[:, :, rows, cols, :h3]
    ref = s0.double() + torch.fft.irfftn(kept, s=dout)
    t1, t2, m3 = plan
    scale = 1.0 / (dout[0] * dout[1] * dout[2])
    td = t.to(dev())

    def run():
        a = _native.fft_resample3d(td, dout, (t1, t1), (t2, t2), m3, scale, adjoint=False, out=s0.to(dev()))
        b, act = _native.fft_resample3d(td, dout, (t1, t1), (t2, t2), m3, scale, adjoint=False, out=s0.to(dev()), act=True)
        return a, b, act
    (a, b, act), ran = _run_profiled(run)
    assert rel_err(a.cpu().numpy(), ref.numpy()) < TOL and torch.equal(a, b)
    assert rel_err(act.cpu().numpy(), torch.nn.functional.gelu(ref).numpy()) < TOL
    COVERED.update(n for n in ran if _spectral(n))


# ------------------------------------------------------------------ C5: 1024^2 block, batch 4, f32 (the mixed form: tests/test_hip_c5.py)
def test_c5_block_f32_bench_batch():
    _check2d(4, 64, 64, 1024, 1024, 1024, 1024, 32, 32, seed=1024)


# C5 model, f32: UNO_9(3, 64, pad=5) at 1024^2 (padded grid 1089), batch 4 - the transforms of its levels (1089 / 544 / 272)
D5 = 1089
C5_LAYERS = {"conv0": (64, 128, D5, D5 // 2, 18), "conv1": (128, 256, D5 // 2, D5 // 4, 8), "conv4": (256, 128, D5 // 4, D5 // 2, 8),
             "conv5": (128, 64, D5 // 2, D5, 18)}


@pytest.mark.parametrize("layer", list(C5_LAYERS))
def test_c5_model_layers_full_size(layer):
    Ci, Co, H, Ho, m = C5_LAYERS[layer]
    _check2d(4, Ci, Co, H, H, Ho, Ho, m, m, seed=5000 + H)


def _mixed(name):
    """mixed-precision kernels (bf16 images / fp16 weights): their full-size comparisons are tests/test_hip_c5.py and
    tests/test_hip_b16_transforms.py (1024^2 and 1089-column cases against the float64 oracle and the float32 reference)"""
    return "b16" in name or "bf16" in name or "__hip_bfloat16" in name or re.search(r"mode_gemm_blocks_kernel<\d+, \d+, \d+, true", name) \
        or re.search(r"mode_gemm_kernel<\d+, (true|false), true", name)


# ------------------------------------------------------------------ the workloads' own launch records
def test_zz_every_bench_table_kernel_was_oracle_checked():
    """Every spectral-path kernel that ONE step / call of any workload bench.py times launches - read off the library's launch
    records of a run made HERE (bench.workload_kernel_names), not from a committed file - also ran inside a full-size oracle
    comparison of this module."""
    if len(COVERED) < 10:
        pytest.skip("the full-size parity tests of this module did not run in this session")
    import bench
    census = bench.workload_kernel_names(torch.device("cuda:0"))
    covered = {n.replace("uno::", "") for n in COVERED}
    missing = {}
    for wl, names in census.items():
        miss = sorted({n.replace("uno::", "") for n in names if _spectral(n) and not _mixed(n)} - covered)
        if miss:
            missing[wl] = miss
    assert not missing, f"spectral kernels of bench workloads that no full-size oracle comparison reached: {missing}"
