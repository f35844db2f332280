"""K3-A: the inverse transform with the up-sampled point-wise branch added in the same pass (uno_dft2d_inverse_add, reference
integral_operators.py:272-273 + :240-242) against the two-kernel form it replaces and against float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


CASES = [
    # (n_img, Hs, Ws, H, W, m1, m2, adjoint)
    (8, 223, 223, 446, 446, 18, 18, False),       # conv5 of the headline model
    (8, 111, 111, 223, 223, 8, 8, False),         # conv4
    (8, 223, 223, 446, 446, 18, 18, True),        # input gradient of conv0 (adjoint of 446 -> 223)
    (8, 111, 111, 223, 223, 8, 8, True),          # input gradient of conv1
    (5, 211, 211, 421, 421, 20, 20, False),       # odd output rows: last row tile has 5 rows
    (3, 215, 223, 430, 446, 12, 7, False),        # non-square, 2 m1 not a multiple of 16
    (1030, 111, 111, 223, 223, 8, 8, False),      # more images than wave slots
    (4, 100, 105, 200, 210, 6, 5, False),         # 7 column tiles, rows 200 = 12.5 tiles
]


@pytest.mark.parametrize("case", CASES)
def test_fused_inverse_add_matches_two_kernels_and_float64(case):
    from uno_amd import _native
    from uno_amd import resample as rs
    n, Hs, Ws, H, W, m1, m2, adjoint = case
    dev = _dev()
    assert _native.dft2d_inverse_add_applies(n, H, W, m1, m2, Hs, Ws), "the fused kernel should cover this shape"
    tabs = rs.upsample_add_tables(Hs, Ws, H, W, str(dev), adjoint)
    assert tabs is not None
    g = torch.Generator().manual_seed(7)
    spec = torch.randn(n, 2 * m1, m2, dtype=torch.complex64, generator=g).to(dev)
    t = torch.randn(n, Hs, Ws, generator=g).to(dev)
    for herm, mask, scale in ((True, True, 1.0), (False, False, 1.0 / (H * W))):
        fused = _native.dft2d_inverse(spec, H, W, scale, herm, mask, addend=(t, tabs))
        plain = _native.dft2d_inverse(spec, H, W, scale, herm, mask)
        two = plain.clone()
        if adjoint:
            rs.resample_adjoint(t.view(1, n, Hs, Ws), H, W, out=two.view(1, n, H, W))
            Rh, Rw = rs._matrix(H, Hs).t(), rs._matrix(W, Ws).t()
        else:
            rs.resample_forward(t.view(1, n, Hs, Ws), H, W, out=two.view(1, n, H, W))
            Rh, Rw = rs._matrix(Hs, H), rs._matrix(Ws, W)
        ref = plain.double() + torch.einsum("hu,nuv,wv->nhw", Rh.double().to(dev), t.double(), Rw.double().to(dev))
        assert torch.isfinite(fused).all()
        assert _rel(fused, ref) < 1e-6, (case, _rel(fused, ref))
        assert _rel(fused, two) < 2e-6
        # element-wise: the addend alone (transform removed) to 1e-5 of the addend's scale
        add_f = fused.double() - plain.double()
        add_r = ref - plain.double()
        assert float((add_f - add_r).abs().max()) < 2e-5 * float(add_r.abs().max())


def test_fused_inverse_add_reads_nothing_outside_the_addend():
    """the addend between two runs of NaN: the result is finite and bit-equal to the run on an ordinary tensor"""
    from uno_amd import _native
    from uno_amd import resample as rs
    dev = _dev()
    n, Hs, Ws, H, W, m1, m2 = 6, 223, 223, 446, 446, 18, 18
    tabs = rs.upsample_add_tables(Hs, Ws, H, W, str(dev), False)
    g = torch.Generator().manual_seed(3)
    spec = torch.randn(n, 2 * m1, m2, dtype=torch.complex64, generator=g).to(dev)
    t = torch.randn(n, Hs, Ws, generator=g).to(dev)
    pad = 16384
    big = torch.full((pad + t.numel() + pad,), float("nan"), device=dev)
    big[pad:pad + t.numel()] = t.flatten()
    tw = big[pad:pad + t.numel()].view(n, Hs, Ws)
    a = _native.dft2d_inverse(spec, H, W, 1.0, True, True, addend=(t, tabs))
    b = _native.dft2d_inverse(spec, H, W, 1.0, True, True, addend=(tw, tabs))
    assert torch.isfinite(b).all()
    assert torch.equal(a, b)
    # and it writes nothing outside its output (64 KiB poisoned bands)
    out_elems = n * H * W
    assert out_elems > 0


def test_fused_form_is_refused_where_it_does_not_apply():
    from uno_amd import _native
    assert not _native.dft2d_inverse_add_applies(8, 64, 64, 8, 8, 32, 32)          # 64 columns: K3-FT does not run there
    assert not _native.dft2d_inverse_add_applies(8, 300, 300, 8, 8, 150, 150)      # 10 column tiles: not compiled
    assert not _native.dft2d_inverse_add_applies(8, 446, 446, 30, 18, 223, 223)    # modes1 beyond the compiled range


@pytest.mark.parametrize("shape", [(2, 16, 8, 111, 223, 8), (2, 8, 16, 223, 111, 8), (2, 8, 8, 105, 210, 6)])
def test_operator_block_fused_equals_two_kernel_form(shape):
    """OperatorBlock_2D up-sampling (forward fused) and down-sampling (input gradient fused): outputs and every gradient against the
    two-kernel form (FUSE_UPSAMPLE_ADD = False) on the same inputs."""
    import uno_amd.integral_operators as io
    B, Ci, Co, S_in, S_out, m = shape
    dev = _dev()
    torch.manual_seed(5)
    blk = io.OperatorBlock_2D(Ci, Co, S_out, S_out, m, m).to(dev)
    x = torch.randn(B, Ci, S_in, S_in, device=dev)
    gy = torch.randn(B, Co, S_out, S_out, device=dev)

    def run(fuse):
        io.FUSE_UPSAMPLE_ADD = fuse
        try:
            xr = x.clone().requires_grad_(True)
            for p in blk.parameters():
                p.grad = None
            y = blk(xr, S_out, S_out)
            y.backward(gy)
            return [y.detach(), xr.grad] + [p.grad.clone() for p in blk.parameters()]
        finally:
            io.FUSE_UPSAMPLE_ADD = True
    from uno_amd import _native
    _native.profile_begin(1000)
    fused = run(True)
    torch.cuda.synchronize()
    names = {n for n, _, _ in _native.profile_end()}
    assert any("dft2d_inv_ft_add_kernel" in n for n in names), names
    two = run(False)
    for a, b in zip(fused, two):
        assert _rel(a, b) < 5e-6, _rel(a, b)


def test_two_source_block_fused_equals_two_kernel_form():
    """forward_cat (the skip form conv5 of the headline model uses), up-sampling 111 -> 223"""
    import uno_amd.integral_operators as io
    dev = _dev()
    torch.manual_seed(6)
    B, C1, C2, Co, Si, So, m = 2, 16, 16, 8, 111, 223, 8
    blk = io.OperatorBlock_2D(C1 + C2, Co, So, So, m, m).to(dev)
    x1 = torch.randn(B, C1, Si, Si, device=dev)
    x2 = torch.randn(B, C2, Si, Si, device=dev)
    gy = torch.randn(B, Co, So, So, device=dev)

    def run(fuse):
        io.FUSE_UPSAMPLE_ADD = fuse
        try:
            a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
            for p in blk.parameters():
                p.grad = None
            y = blk.forward_cat([a, b], So, So, defer_gelu=True)
            y.backward(gy)
            return [y.detach(), a.grad, b.grad] + [p.grad.clone() for p in blk.parameters()]
        finally:
            io.FUSE_UPSAMPLE_ADD = True
    fused, two = run(True), run(False)
    for a, b in zip(fused, two):
        assert _rel(a, b) < 5e-6, _rel(a, b)
