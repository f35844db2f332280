"""Mixed-precision form of the 2-D spectral convolution (BASELINE.json config 5: bf16 activations, half-precision weight
storage, f32 accumulation) through the C ABI.  Needs a real MI355X:  pytest -m gpu

The bf16 kernels are the f32 kernels with a widening load (<< 16) / a round-to-nearest-even store, so the contract is exact:
  result_bf16(x_bf16) == round_bf16(result_f32(widen(x_bf16)))   bit for bit   (torch's .bfloat16() is the same RNE)
plus one comparison against the float64 oracle at the tolerance bf16 output rounding allows
  TOL_BF16 = 3e-3 relative L2  (half an ulp of bf16 is 2^-9 = 2e-3 relative per element)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
TOL = 2e-5
TOL_BF16 = 3e-3


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


SHAPES = [  # n_img, H, W, m1, m2  (odd widths: 2-byte aligned 8-byte loads / stores)
    (3, 16, 16, 4, 5), (2, 21, 18, 4, 5), (2, 23, 23, 11, 12), (1, 40, 44, 17, 20), (2, 85, 85, 12, 12), (1, 7, 130, 3, 40),
    (2, 10, 14, 7, 5), (1, 1, 2, 1, 2), (2, 111, 111, 8, 8), (1, 421, 421, 20, 20), (300, 16, 15, 6, 6), (1, 64, 66, 32, 33),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_dft_reads_bf16(shape):
    from uno_amd import _native
    n, H, W, m1, m2 = shape
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(n, 1, H, W, generator=g).bfloat16().to(dev())
    got = _native.dft2d_forward(x, m1, m2, scale=0.5, hermitian_cols=True, mask_overlap=True)
    want = _native.dft2d_forward(x.float(), m1, m2, scale=0.5, hermitian_cols=True, mask_overlap=True)
    # same arithmetic on the same (widened) values; the f32 path may take the plane-batched kernel, which sums in another order
    assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < TOL
    ref = so.truncated_rfft2_dense(x.float().cpu().numpy(), m1, m2) * (H * W) * 0.5
    ref = ref * so.hermitian_weights(W, m2)[None, None, None, :] * so.later_wins_mask(H, m1)[None, None, :, None]
    assert rel_err(got.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("shape", SHAPES)
def test_inverse_dft_writes_bf16_rounded_to_nearest_even(shape):
    from uno_amd import _native
    n, H, W, m1, m2 = shape
    g = torch.Generator().manual_seed(H * 17 + W)
    O = torch.randn(n, 1, 2 * m1, m2, dtype=torch.cfloat, generator=g).to(dev())
    got = _native.dft2d_inverse(O, H, W, scale=0.25, dtype=torch.bfloat16)
    want = _native.dft2d_inverse(O, H, W, scale=0.25)
    assert got.dtype == torch.bfloat16 and got.shape == want.shape
    NT, MT = (m2 + 15) // 16, (2 * m1 + 15) // 16
    b16 = W >= 64 and H >= 16 and m2 <= 32 and NT * MT <= 8          # these run the bf16-MFMA form (csrc/dft2d_b16.hip, tests/test_hip_b16_transforms.py)
    if n < 128 and not b16:     # same kernel, f32 values identical before the store
        assert torch.equal(got, want.bfloat16())
    else:                # another kernel form (plane-batched / bf16 MFMA: other summation order): equal up to one bf16 ulp
        assert rel_err(got.float().cpu().numpy(), want.cpu().numpy()) < TOL_BF16


@pytest.mark.parametrize("cfg", [(2, 3, 4, 21, 18, 13, 10, 4, 5), (2, 8, 6, 40, 44, 64, 60, 12, 14), (1, 4, 4, 85, 85, 43, 43, 12, 12)])
def test_operator_forward_backward_equal_the_f32_operator_on_widened_inputs(cfg):
    from uno_amd.integral_operators import spectral_conv2d, spectral_conv2d_mixed
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = cfg
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Ci, H, W, generator=g).bfloat16().to(dev())
    w1 = (0.3 * torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g)).to(dev())
    w2 = (0.3 * torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g)).to(dev())
    gy = torch.randn(B, Co, Ho, Wo, generator=g).bfloat16().to(dev())

    xm, w1m, w2m = x.clone().requires_grad_(), w1.clone().requires_grad_(), w2.clone().requires_grad_()
    ym = spectral_conv2d_mixed(xm, w1m, w2m, Ho, Wo)
    ym.backward(gy)
    xf, w1f, w2f = x.float().requires_grad_(), w1.clone().requires_grad_(), w2.clone().requires_grad_()
    yf = spectral_conv2d(xf, w1f, w2f, Ho, Wo)
    yf.backward(gy.float())
    assert ym.dtype == torch.bfloat16 and xm.grad.dtype == torch.bfloat16 and w1m.grad.dtype == torch.complex64
    # the f32 call may take another kernel form than the bf16 call (full-tile / half-tile forms exist for f32 images only): the
    # same values summed in another order, i.e. equal up to one bf16 ulp of the output and to f32 rounding in the weight gradients
    assert rel_err(ym.detach().float().cpu().numpy(), yf.detach().bfloat16().float().cpu().numpy()) < TOL_BF16
    assert rel_err(xm.grad.float().cpu().numpy(), xf.grad.bfloat16().float().cpu().numpy()) < TOL_BF16
    assert rel_err(w1m.grad.cpu().numpy(), w1f.grad.cpu().numpy()) < TOL and rel_err(w2m.grad.cpu().numpy(), w2f.grad.cpu().numpy()) < TOL
    # and against the float64 oracle of the reference operator on the widened inputs
    ref = so.spectral_conv2d_dense(x.float().cpu().numpy(), w1.cpu().numpy(), w2.cpu().numpy(), Ho, Wo)[0]
    assert rel_err(ym.detach().float().cpu().numpy(), ref) < TOL_BF16


def test_half_precision_weight_storage_and_argument_errors():
    from uno_amd.integral_operators import SpectralConv2d_Uno, spectral_conv2d_mixed
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 20, 20, generator=g).bfloat16().to(dev())
    w = [(0.3 * torch.randn(3, 5, 4, 4, dtype=torch.cfloat, generator=g)).to(dev()) for _ in range(2)]
    wh = [torch.view_as_real(t).half().requires_grad_() for t in w]
    y = spectral_conv2d_mixed(x, wh[0], wh[1], 16, 16)
    want = spectral_conv2d_mixed(x, torch.view_as_complex(wh[0].detach().float()), torch.view_as_complex(wh[1].detach().float()), 16, 16)
    assert torch.equal(y, want)
    y.float().sum().backward()
    assert wh[0].grad.dtype == torch.float16 and wh[0].grad.shape == wh[0].shape and bool(torch.isfinite(wh[0].grad.float()).all())
    with pytest.raises(RuntimeError):
        spectral_conv2d_mixed(x.float(), w[0], w[1], 16, 16)            # the mixed form takes bf16 activations only
    with pytest.raises(RuntimeError):
        SpectralConv2d_Uno(3, 5, 16, 16, 4, 4).to(dev())(x)             # the reference's module contract: float32 input


def test_mixed_precision_layer_with_default_modes_widens_instead_of_raising():
    """enable_mixed_precision on a layer built with the reference's DEFAULT modes (dim1//2 - 1, dim2//2: beyond the bf16 kernels'
    compiled range): the layer runs its float32 any-mode form on the widened activations and returns bf16 - same result as the
    f32 layer on the widened input, rounded once."""
    from uno_amd.integral_operators import OperatorBlock_2D, SpectralConv2d_Uno, enable_mixed_precision
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    conv = SpectralConv2d_Uno(3, 2, 100, 100).to(dev)
    assert (conv.modes1, conv.modes2) == (49, 50)
    xb = torch.randn(2, 3, 100, 100, device=dev).bfloat16()
    ref = conv(xb.float())
    enable_mixed_precision(conv)
    y = conv(xb)
    assert y.dtype == torch.bfloat16
    assert torch.equal(y, ref.bfloat16())
    blk = enable_mixed_precision(OperatorBlock_2D(3, 2, 100, 100, 49, 50).to(dev))
    out = blk(xb)
    assert out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    out.float().sum().backward()
    assert blk.conv.weights1.grad is not None and torch.isfinite(torch.view_as_real(blk.conv.weights1.grad)).all()


def test_half_weight_copy_is_recorded_in_a_captured_graph():
    """The float16 (re, im) copy of the complex64 master weights is cached per parameter version - but while a HIP graph is being
    captured the conversion must be part of the graph: the optimiser updates the master weights BETWEEN replays (eagerly), and a copy
    found in the cache at capture time would be frozen into every replay."""
    from uno_amd.integral_operators import SpectralConv2d_Uno, enable_mixed_precision
    torch.manual_seed(0)
    layer = enable_mixed_precision(SpectralConv2d_Uno(8, 8, 24, 24, 5, 5).to(dev()))
    x = torch.randn(2, 8, 24, 24, device=dev()).bfloat16()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        layer(x)                                        # eager warm-up: fills the per-parameter cache
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        y = layer(x)
    with torch.no_grad():
        layer.weights1.mul_(-1.5)                       # what an optimiser step does: in-place update of the master weights
        layer.weights2.add_(0.25)
    graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        want = layer(x)
    assert torch.equal(y, want), "the replay read a stale half-precision weight copy"
