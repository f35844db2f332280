"""Pins the CPU oracle (oracle/spectral_oracle.py) against golden vectors produced by
the genuine reference (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import Case, load_cases, rel_err
from oracle import spectral_oracle as so

Z2, NAMES2 = load_cases("spectral2d.npz")
Z3, NAMES3 = load_cases("spectral3d.npz")
ZB, NAMESB = load_cases("blocks.npz")
CASES2 = [n for n in NAMES2 if n not in ("fp64_in",)]

TOL_DENSE = 5e-6      # float64 dense DFT vs the reference's float32 FFT path
TOL_FFT = 2e-6        # same op sequence, float32


@pytest.mark.parametrize("name", CASES2)
def test_dense2d_forward_backward(name):
    c = Case(Z2, name)
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = [int(v) for v in c.meta]
    y, X = so.spectral_conv2d_dense(c.x, c.w1, c.w2, Ho, Wo)
    assert y.shape == c.y.shape
    assert rel_err(y, c.y) < TOL_DENSE
    gx, gw1, gw2, _, _ = so.spectral_conv2d_dense_bwd(c.gy, X, c.w1, c.w2, H, W)
    assert rel_err(gx, c.gx) < TOL_DENSE
    assert rel_err(gw1, c.gw1) < TOL_DENSE
    assert rel_err(gw2, c.gw2) < TOL_DENSE
    # overlap: rows of weights1 overwritten by the later corner receive exactly zero grad
    zero_ref = (c.gw1 == 0)
    if zero_ref.any():
        assert np.all(np.abs(gw1[zero_ref]) == 0)


@pytest.mark.parametrize("name", CASES2)
def test_fft2d_forward_backward(name):
    c = Case(Z2, name)
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = [int(v) for v in c.meta]
    x = torch.from_numpy(c.x).requires_grad_(True)
    w1 = torch.from_numpy(c.w1).requires_grad_(True)
    w2 = torch.from_numpy(c.w2).requires_grad_(True)
    y = so.spectral_conv2d_fft(x, w1, w2, Ho, Wo)
    assert y.dtype == torch.float32
    assert rel_err(y.detach().numpy(), c.y) < TOL_FFT
    y.backward(torch.from_numpy(c.gy))
    assert rel_err(x.grad.numpy(), c.gx) < TOL_FFT
    assert rel_err(w1.grad.numpy(), c.gw1) < TOL_FFT
    assert rel_err(w2.grad.numpy(), c.gw2) < TOL_FFT


def test_fp64_input_is_rejected_by_reference():
    assert int(Z2["fp64_in.raises"]) == 1


@pytest.mark.parametrize("name", NAMES3)
def test_dense3d_forward_backward(name):
    c = Case(Z3, name)
    meta = [int(v) for v in c.meta]
    B, Ci, Co = meta[:3]
    din, dout, modes = meta[3:6], meta[6:9], meta[9:12]
    ws = [getattr(c, f"w{k}") for k in range(1, 5)]
    y, X = so.spectral_conv3d_dense(c.x, ws, *dout)
    assert rel_err(y, c.y) < TOL_DENSE
    gx, gws, _, _ = so.spectral_conv3d_dense_bwd(c.gy, X, ws, *din)
    assert rel_err(gx, c.gx) < TOL_DENSE
    for k in range(4):
        ref = getattr(c, f"gw{k + 1}")
        assert rel_err(gws[k], ref) < TOL_DENSE
        assert np.all(gws[k][ref == 0] == 0)


@pytest.mark.parametrize("name", NAMES3)
def test_fft3d_forward(name):
    c = Case(Z3, name)
    meta = [int(v) for v in c.meta]
    dout = meta[6:9]
    ws = [torch.from_numpy(getattr(c, f"w{k}")) for k in range(1, 5)]
    y = so.spectral_conv3d_fft(torch.from_numpy(c.x), ws, *dout)
    assert rel_err(y.numpy(), c.y) < TOL_FFT


def test_overlap_zero_grad_fractions_3d():
    """zero-grad fractions 0.889/0.667/0.667/0 for weights1..4 at
    (16,16,20 -> 8,8,20; m=6,6,7)."""
    c = Case(Z3, "overlap_T40conv3")
    fr = [float((getattr(c, f"gw{k}") == 0).mean()) for k in range(1, 5)]
    assert np.allclose(fr, [8 / 9, 2 / 3, 2 / 3, 0.0], atol=1e-3)


B2D = [n for n in NAMESB if n.startswith("b2d_")]
B3D = [n for n in NAMESB if n.startswith("b3d_")]
PW3D = [n for n in NAMESB if n.startswith("pw3d_")]


def _load_sd(mod, sd):
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)


@pytest.mark.parametrize("name", B2D)
def test_block2d(name):
    c = Case(ZB, name)
    B, Ci, Co, H, W, Ho, Wo, m1, m2, nrm, nl = [int(v) for v in c.meta]
    blk = so.OracleOperatorBlock2d(Ci, Co, Ho, Wo, m1, m2, Normalize=bool(nrm), Non_Lin=bool(nl))
    _load_sd(blk, c.sub("sd"))
    x = torch.from_numpy(c.x).requires_grad_(True)
    y = blk(x)
    assert rel_err(y.detach().numpy(), c.y) < 1e-5
    y.backward(torch.from_numpy(c.gy))
    assert rel_err(x.grad.numpy(), c.gx) < 1e-5
    for k, g in c.sub("grad").items():
        assert rel_err(dict(blk.named_parameters())[k].grad.numpy(), g) < 2e-5, k


@pytest.mark.parametrize("name", B3D)
def test_block3d(name):
    c = Case(ZB, name)
    meta = [int(v) for v in c.meta]
    B, Ci, Co = meta[:3]
    dout, modes, nrm, nl = meta[6:9], meta[9:12], meta[12], meta[13]
    blk = so.OracleOperatorBlock3d(Ci, Co, *dout, *modes, Normalize=bool(nrm), Non_Lin=bool(nl))
    _load_sd(blk, c.sub("sd"))
    x = torch.from_numpy(c.x).requires_grad_(True)
    y = blk(x)
    assert rel_err(y.detach().numpy(), c.y) < 1e-5
    y.backward(torch.from_numpy(c.gy))
    assert rel_err(x.grad.numpy(), c.gx) < 2e-5


@pytest.mark.parametrize("name", PW3D)
def test_pointwise3d_bug_compatible(name):
    c = Case(ZB, name)
    meta = [int(v) for v in c.meta]
    dout = meta[6:9]
    y = so.pointwise3d(torch.from_numpy(c.x), torch.from_numpy(c.weight), torch.from_numpy(c.bias), *dout)
    assert rel_err(y.numpy(), c.y) < 1e-5


def test_dim_mutation():
    c = Case(ZB, "dimmut")
    blk = so.OracleOperatorBlock2d(3, 4, 16, 16, 4, 4)
    _load_sd(blk, c.sub("sd"))
    x = torch.from_numpy(c.x)
    y = blk(x, 12, 12)
    assert rel_err(y.detach().numpy(), c.y_override) < 1e-5
    assert [blk.conv.dim1, blk.conv.dim2, blk.w.dim1, blk.w.dim2] == [int(v) for v in c.state]
    assert rel_err(blk.conv(x).detach().numpy(), c.y_conv_after) < 1e-5


def test_reference_adam_and_loss():
    z, _ = load_cases("harness.npz")
    c = Case(z, "adam")
    pc = torch.from_numpy(c.pc0.copy())
    pr = torch.from_numpy(c.pr0.copy())
    ms = [torch.zeros_like(pc), torch.zeros_like(pr)]
    vs = [torch.zeros_like(pc), torch.zeros_like(pr)]
    for t in range(3):
        so.reference_adam_step([pc, pr], [torch.from_numpy(c.gc[t]), torch.from_numpy(c.gr[t])], ms, vs,
                               t + 1, 1e-2, 0.9, 0.999, 1e-8, 1e-3)
    assert rel_err(pc.numpy(), c.pc3) < 1e-6
    assert rel_err(pr.numpy(), c.pr3) < 1e-6
