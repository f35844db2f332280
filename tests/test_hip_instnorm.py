"""K13 (csrc/instnorm.hip): InstanceNorm(affine) [+ GELU] of the operator blocks (reference integral_operators.py:269-270,
277-283) against the same modules evaluated in float64 on the CPU - forward, input gradient, weight / bias gradients.
Tolerance: 3e-6 l2-relative (two-pass statistics in f32; MIOpen's path is 3e-4 off on odd grids)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    d = (a.double().cpu() - b).norm().item()
    n = b.norm().item()
    return d / n if n > 0 else d


# rows of 35 .. 64000 floats run the register-resident kernels (4, 8, 13, 16, 25, 32 quads per thread), 90000 the sweep kernels
@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (1, 1, 1, 1), (3, 16, 111, 111), (2, 8, 223, 223), (2, 4, 12, 10, 9), (2, 6, 1003),
                                   (1, 3, 150, 150), (1, 2, 180, 181), (1, 2, 256, 250), (1, 2, 300, 300), (2, 2, 3)])
@pytest.mark.parametrize("gelu", [True, False])
@pytest.mark.parametrize("affine", [True, False])
def test_instance_norm_gelu(shape, gelu, affine):
    from uno_amd.integral_operators import instance_norm_gelu
    nd = len(shape) - 2
    cls = {1: nn.InstanceNorm1d, 2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}[nd]
    torch.manual_seed(sum(shape))
    norm = cls(shape[1], affine=affine)
    if affine:
        with torch.no_grad():
            norm.weight.copy_(torch.randn(shape[1]))
            norm.bias.copy_(torch.randn(shape[1]))
    x = (3.0 * torch.randn(*shape) + 1.5)
    gy = torch.randn(*shape)

    ref_norm = cls(shape[1], affine=affine).double()
    ref_norm.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    if shape[2:] == (1, 1):          # a single point per row: torch raises, so does the fused path
        from uno_amd.integral_operators import instance_norm_gelu as ing
        with pytest.raises(ValueError):
            ref_norm(x.double())
        with pytest.raises(ValueError):
            ing(x.cuda(), norm.cuda(), gelu)
        return
    x2 = x.double().requires_grad_(True)
    y2 = ref_norm(x2)
    if gelu:
        y2 = F.gelu(y2)
    ref = torch.autograd.grad(y2, [x2] + list(ref_norm.parameters()), gy.double())

    norm = norm.cuda()
    xd = x.cuda().requires_grad_(True)
    y = instance_norm_gelu(xd, norm, gelu)
    got = torch.autograd.grad(y, [xd] + list(norm.parameters()), gy.cuda())
    assert rel(y, y2.detach()) < 3e-6
    assert rel(got[0], ref[0]) < 2e-5
    for a, r in zip(got[1:], ref[1:]):
        assert a.shape == r.shape and rel(a, r) < 2e-5


def test_block_uses_fused_norm_and_matches_stock_modules():
    from uno_amd.integral_operators import OperatorBlock_2D
    torch.manual_seed(1)
    blk = OperatorBlock_2D(4, 6, 21, 19, 3, 3, Normalize=True).cuda()
    x = torch.randn(2, 4, 30, 27, device="cuda")
    y = blk(x)
    s = blk.conv(x) + blk.w(x)
    y2 = F.gelu(F.instance_norm(s.double(), weight=blk.normalize_layer.weight.double(), bias=blk.normalize_layer.bias.double(), eps=1e-5))
    assert rel(y, y2.detach().cpu()) < 1e-5
