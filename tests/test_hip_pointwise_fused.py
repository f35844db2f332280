"""K11 / K12 (csrc/pointwise_fused.hip) against the stock op sequences they replace: F.gelu -> Linear(C, 1) (reference
darcy_flow_uno2d.py:128-131) and F.gelu -> F.pad (darcy_flow_uno2d.py:103-107), forward and all gradients, in float64
on the device.  Tolerance 2e-6 (l2-relative) forward, 2e-5 for the pixel-long weight / bias reductions."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    d = (a.double() - b).norm().item()
    n = b.norm().item()
    return d / n if n > 0 else d


@pytest.mark.parametrize("B,C,shape", [(1, 1, (1,)), (2, 5, (3, 7)), (3, 64, (45, 41)), (2, 128, (4099,)), (2, 20, (6, 5, 7))])
@pytest.mark.parametrize("with_bias", [True, False])
def test_gelu_project(B, C, shape, with_bias):
    from uno_amd.integral_operators import gelu_project
    g = torch.Generator().manual_seed(B + C)
    pre = (2.0 * torch.randn(B, C, *shape, generator=g)).cuda().requires_grad_(True)
    w = torch.randn(1, C, generator=g).cuda().requires_grad_(True)
    b = torch.randn(1, generator=g).cuda().requires_grad_(True) if with_bias else None
    y = gelu_project(pre, w, b)
    assert y.shape == (B, 1, *shape)
    gy = torch.randn_like(y)
    params = (pre, w) + ((b,) if with_bias else ())
    got = torch.autograd.grad(y, params, gy)
    pre2, w2 = pre.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    b2 = b.detach().double().requires_grad_(True) if with_bias else None
    y2 = torch.einsum("oc,bc...->bo...", w2, F.gelu(pre2))
    if with_bias:
        y2 = y2 + b2.view(1, 1, *([1] * len(shape)))
    ref = torch.autograd.grad(y2, (pre2, w2) + ((b2,) if with_bias else ()), gy.double())
    assert rel(y, y2.detach()) < 2e-6
    assert rel(got[0], ref[0]) < 2e-6
    for a, r in zip(got[1:], ref[1:]):
        assert a.shape == r.shape and rel(a, r) < 2e-5


@pytest.mark.parametrize("lead,hw,pad", [((2, 3), (5, 7), (2, 3)), ((1, 1), (1, 1), (0, 0)), ((2, 8), (41, 37), (5, 5)), ((3,), (16, 130), (0, 9))])
def test_gelu_pad(lead, hw, pad):
    from uno_amd.integral_operators import gelu_pad2d
    g = torch.Generator().manual_seed(sum(hw))
    s = (2.0 * torch.randn(*lead, *hw, generator=g)).cuda().requires_grad_(True)
    y = gelu_pad2d(s, pad[0], pad[1])
    assert y.shape == (*lead, hw[0] + pad[0], hw[1] + pad[1])
    gy = torch.randn_like(y)
    (gs,) = torch.autograd.grad(y, s, gy)
    s2 = s.detach().double().requires_grad_(True)
    y2 = F.pad(F.gelu(s2), [0, pad[1], 0, pad[0]])
    (gs2,) = torch.autograd.grad(y2, s2, gy.double())
    assert rel(y, y2.detach()) < 2e-6
    assert torch.equal(y[..., hw[0]:, :], torch.zeros_like(y[..., hw[0]:, :])) and torch.equal(y[..., :, hw[1]:], torch.zeros_like(y[..., :, hw[1]:]))
    assert rel(gs, gs2) < 2e-6


def test_wgrad_is_reproducible():
    from uno_amd import _native
    pre = torch.randn(4, 64, 9000, device="cuda")
    w = torch.randn(64, device="cuda")
    go = torch.randn(4, 9000, device="cuda")
    a = _native.gelu_project_backward(pre, w, go)
    b = _native.gelu_project_backward(pre, w, go)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_device_gelu_and_its_derivative_against_float64():
    """The kernels' exact-erf GELU (uno_common.h: one branch-free rational erf, v_exp for the density) and gelu' on 2 M points - a dense
    grid over [-8, 8] plus N(0, 4) samples - against float64 (reference: F.gelu, integral_operators.py:282 and its autograd).  Measured
    2.4e-7 max(1, |x|) / 2.5e-7 (torch's own float32 kernels: 4.5e-7 / 1.4e-7 absolute)."""
    from uno_amd import _native
    n = 1 << 20
    x = torch.cat([torch.linspace(-8, 8, n, dtype=torch.float64), torch.randn(n, dtype=torch.float64) * 2]).float()
    H, W = 2048, 1024
    xs = x.view(1, H, W).cuda()
    g = _native.gelu_pad(xs, H, W).double()
    d = _native.gelu_pad_backward(xs, torch.ones_like(xs)).double()
    xd = xs.double()
    Phi = 0.5 * (1 + torch.erf(xd / 2 ** 0.5))
    phi = torch.exp(-0.5 * xd * xd) / (2 * torch.pi) ** 0.5
    assert float(((g - xd * Phi).abs() / xd.abs().clamp(min=1)).max()) < 4e-7
    assert float((d - (Phi + xd * phi)).abs().max()) < 5e-7
    # the tails: gelu(x) -> x and 0, gelu'(x) -> 1 and 0, no NaN from the clamped argument (the clamped erf leaves Phi(-inf) = 3.9e-9
    # instead of 0: |gelu(x)| <= 4e-9 |x| on the far negative side)
    far = torch.tensor([-1e4, -50.0, -9.0, 9.0, 50.0, 1e4]).view(1, 1, 6).cuda()
    gf, df = _native.gelu_pad(far, 1, 6).flatten(), _native.gelu_pad_backward(far, torch.ones_like(far)).flatten()
    assert torch.equal(gf[3:], far.flatten()[3:]) and bool((gf[:3].abs() <= 1e-8 * far.flatten()[:3].abs()).all())
    assert float((df[3:] - 1).abs().max()) < 1e-6 and float(df[:3].abs().max()) < 1e-6
