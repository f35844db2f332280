"""HIP separable resampling (pointwise_op_2D's bicubic anti-aliased resize) vs torch's CPU op.  pytest -m gpu"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

SIZES = [((20, 18), (12, 10)), ((12, 10), (21, 19)), ((90, 90), (45, 45)), ((45, 45), (22, 22)), ((22, 22), (45, 45)),
         ((45, 45), (90, 90)), ((37, 50), (29, 31)), ((223, 223), (111, 111)), ((111, 111), (223, 223)), ((16, 16), (16, 16)),
         ((9, 300), (4, 301)), ((446, 446), (223, 223)), ((7, 3), (5, 2)), ((5, 2), (9, 3)), ((33, 5), (20, 7)),
         # rows whose 16 x W tile alone fits 64 KB of LDS but tile + row-operator table does not (W ~ 980..1024: an unpadded
         # 1024^2 level resampled 1024 -> 512): must take the two-pass form, not fail at launch
         ((40, 1024), (20, 512)), ((36, 1000), (18, 500)), ((20, 512), (40, 1024)), ((24, 960), (12, 480)),
         # round 3: rows up to ~2400 floats run the fused kernel with a raised dynamic-LDS limit (the 1089 -> 544 level of config C5);
         # beyond that the two-pass form
         ((33, 1089), (16, 544)), ((16, 544), (33, 1089)), ((6, 2500), (3, 1300)), ((3, 1300), (6, 2500))]


def test_band_tables_reconstruct_the_operator():
    from uno_amd.resample import _band, _matrix
    for n_in, n_out in [(20, 12), (12, 20), (90, 45), (45, 90), (446, 223), (7, 7)]:
        R = _matrix(n_in, n_out)
        for M in (R, R.t().contiguous()):
            s, w, K = _band(M)
            dense = torch.zeros_like(M, dtype=torch.float32)
            for i in range(M.shape[0]):
                for t in range(K):
                    if s[i] + t < M.shape[1]:
                        dense[i, s[i] + t] += w[i, t]
            assert torch.allclose(dense, M.float(), atol=1e-7)
        assert abs(float(R.double().sum(1).sub(1).abs().max())) < 1e-6         # rows sum to 1: a bias commutes with it


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", SIZES)
def test_resample_forward_backward_vs_torch_cpu(sizes):
    from uno_amd.resample import resample2d_bicubic_aa
    (H, W), (Ho, Wo) = sizes
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * 31 + Wo)
    x = torch.randn(2, 3, H, W, generator=g)
    gy = torch.randn(2, 3, Ho, Wo, generator=g)
    xc = x.clone().requires_grad_(True)
    yc = F.interpolate(xc, size=(Ho, Wo), mode="bicubic", align_corners=True, antialias=True)
    yc.backward(gy)
    xd = x.to(dev).requires_grad_(True)
    yd = resample2d_bicubic_aa(xd, Ho, Wo)
    yd.backward(gy.to(dev))
    assert rel_err(yd.detach().cpu().numpy(), yc.detach().numpy()) < 2e-6
    assert rel_err(xd.grad.cpu().numpy(), xc.grad.numpy()) < 2e-6


@pytest.mark.gpu
def test_pointwise_op_commuted_order_matches_reference_order():
    from oracle import spectral_oracle as so
    from uno_amd.integral_operators import pointwise_op_2D
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    for (H, Ho) in [(40, 20), (20, 40)]:
        pw = pointwise_op_2D(5, 7, Ho, Ho)
        x = torch.randn(2, 5, H, H)
        ref = so.pointwise2d(x, pw.conv.weight, pw.conv.bias, Ho, Ho)
        got = pw.to(dev)(x.to(dev))
        assert rel_err(got.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [((40, 36), (20, 18)), ((20, 18), (41, 37)), ((30, 1100), (15, 600)), ((5, 2600), (3, 1300))])
def test_resample_accumulates_into_out(sizes):
    """out= form (fused kernel and the two-pass fallback for rows too long for LDS): out += R x R^T."""
    from uno_amd.resample import resample_forward, resample_adjoint
    (H, W), (Ho, Wo) = sizes
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, H, W, generator=g).to(dev)
    base = torch.randn(2, 3, Ho, Wo, generator=g).to(dev)
    plain = resample_forward(x, Ho, Wo)
    out = base.clone()
    ret = resample_forward(x, Ho, Wo, out=out)
    assert ret.data_ptr() == out.data_ptr()
    assert rel_err(out.cpu().numpy(), (base + plain).cpu().numpy()) < 1e-6
    gbase = torch.randn(2, 3, H, W, generator=g).to(dev)
    gout = gbase.clone()
    resample_adjoint(base, H, W, out=gout)
    assert rel_err(gout.cpu().numpy(), (gbase + resample_adjoint(base, H, W)).cpu().numpy()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [((8, 10, 6), (8, 10, 11)), ((12, 12, 10), (9, 7, 18)), ((6, 5, 4), (6, 5, 4)), ((5, 40, 30), (9, 33, 30)),
                                   ((4, 36, 40), (4, 30, 36))])
def test_trilinear_resample_vs_torch_cpu(sizes):
    """3-D skip-connection resize (reference navier_stokes_uno3d.py:352-372) on the banded kernels vs torch's CPU op."""
    from uno_amd.resample import resample3d_trilinear
    src, dst = sizes
    g = torch.Generator().manual_seed(sum(src) + sum(dst))
    x = torch.randn(2, 3, *src, generator=g)
    gy = torch.randn(2, 3, *dst, generator=g)
    xc = x.clone().requires_grad_(True)
    yc = F.interpolate(xc, size=dst, mode="trilinear", align_corners=True)
    yc.backward(gy)
    xd = x.cuda().requires_grad_(True)
    yd = resample3d_trilinear(xd, dst)
    yd.backward(gy.cuda())
    assert yd.shape == yc.shape
    assert rel_err(yd.detach().cpu().numpy(), yc.detach().numpy()) < 2e-6
    assert rel_err(xd.grad.cpu().numpy(), xc.grad.numpy()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [((40, 40), (28, 28)), ((45, 45), (22, 22)), ((22, 37), (45, 50))])
def test_resample_non_finite_row_stays_in_its_tiles(sizes):
    """An Inf in one input row reaches, in the reference's banded interpolation, only the output rows with a tap on it.  The fused
    kernel applies a dense 16 x NP row operator per tile of 16 output rows, so the row's own tiles may turn non-finite as a whole
     - but no OTHER tile: the rows a tile's last k-step loads beyond its band are replaced by zero, not
    multiplied by a zero weight (round 5; before, an Inf up to three rows past a tile's band made that tile NaN too)."""
    from uno_amd.resample import resample2d_bicubic_aa
    (H, W), (Ho, Wo) = sizes
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(1, 2, H, W, generator=g)
    for r in range(H):
        x = x0.clone()
        x[:, :, r, :] = float("inf")
        ref_bad = ~torch.isfinite(F.interpolate(x, size=(Ho, Wo), mode="bicubic", align_corners=True, antialias=True)).all(dim=(0, 1, 3))   # (Ho,)
        got_bad = ~torch.isfinite(resample2d_bicubic_aa(x.to(dev), Ho, Wo).cpu()).all(dim=(0, 1, 3))
        allowed = torch.zeros(Ho, dtype=torch.bool)
        for i in ref_bad.nonzero().flatten().tolist():
            allowed[(i // 16) * 16:(i // 16) * 16 + 16] = True
        assert bool(ref_bad.any())
        assert not bool((got_bad & ~allowed).any()), f"input row {r}: output rows {(got_bad & ~allowed).nonzero().flatten().tolist()} are non-finite outside the tiles of {ref_bad.nonzero().flatten().tolist()}"
        assert bool((got_bad | ~ref_bad).all()), f"input row {r}: a row the reference makes non-finite came out finite"
