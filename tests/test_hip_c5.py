"""BASELINE.json configs[4] on the GPU at its real grid: Darcy 1024 x 1024, 64 channels, modes 32 (SURVEY.md 8(d) "C5").

The reference has no behaviour in reduced precision (integral_operators.py:187 raises on bf16), so the contract is
  * float32: the product block equals the reference's op sequence (FFT oracle on the host) at TOL = 2e-5 relative L2,
  * mixed (bf16 activations, f32 accumulation, fp16 weight storage): the product equals the float32 REFERENCE applied to the
    pre-rounded inputs, within what one bf16 rounding of the output allows: TOL_BF16 = 3e-3 (half an ulp of bf16 = 2^-9),
plus size-independent properties (determinism, adjoint identity) and a whole-model training step at S = 1024.   pytest -m gpu"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
TOL = 2e-5
TOL_BF16 = 3e-3
C, S, M = 64, 1024, 32


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _block_inputs(B, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, S, S, generator=g)
    sc = (1 / (2 * C)) ** 0.5
    w1 = sc * torch.randn(C, C, M, M, dtype=torch.cfloat, generator=g)
    w2 = sc * torch.randn(C, C, M, M, dtype=torch.cfloat, generator=g)
    gy = torch.randn(B, C, S, S, generator=g)
    return x, w1, w2, gy


def _reference(x, w1, w2, gy):
    xr, w1r, w2r = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    y = so.spectral_conv2d_fft(xr, w1r, w2r, S, S)
    y.backward(gy)
    return y.detach(), xr.grad, w1r.grad, w2r.grad


def test_c5_block_float32_matches_the_reference_sequence():
    """SpectralConv2d(64, 64, 1024, 1024, 32, 32), batch 2: y, gx, gw1, gw2 against rfft2 -> einsum -> irfft2 on the host."""
    from uno_amd.integral_operators import spectral_conv2d
    x, w1, w2, gy = _block_inputs(2)
    y_ref, gx_ref, gw1_ref, gw2_ref = _reference(x, w1, w2, gy)
    xd, w1d, w2d = (t.to(dev()).requires_grad_(True) for t in (x, w1, w2))
    y = spectral_conv2d(xd, w1d, w2d, S, S)
    y.backward(gy.to(dev()))
    assert rel_err(y.detach().cpu().numpy(), y_ref.numpy()) < TOL
    assert rel_err(xd.grad.cpu().numpy(), gx_ref.numpy()) < TOL
    assert rel_err(w1d.grad.cpu().numpy(), gw1_ref.numpy()) < TOL
    assert rel_err(w2d.grad.cpu().numpy(), gw2_ref.numpy()) < TOL
    with torch.no_grad():
        assert torch.equal(spectral_conv2d(xd, w1d, w2d, S, S), y.detach())              # determinism
        a = torch.dot(y.detach().double().flatten(), gy.to(dev()).double().flatten())   # <A x, g> == <x, A^T g>
        b = torch.dot(xd.detach().double().flatten(), xd.grad.double().flatten())
        assert abs(a.item() - b.item()) <= 1e-5 * max(abs(a.item()), abs(b.item()))


def test_c5_block_mixed_matches_float32_reference_on_prerounded_inputs():
    """bf16 activations + fp16 (re, im) weight storage: forward and all gradients against the float32 reference sequence applied
    to exactly the values the kernels see (inputs rounded once to bf16 / fp16)."""
    from uno_amd.integral_operators import spectral_conv2d_mixed
    x, w1, w2, gy = _block_inputs(2, seed=6)
    xb, gyb = x.bfloat16(), gy.bfloat16()
    w1h, w2h = torch.view_as_real(w1).half(), torch.view_as_real(w2).half()
    w1w, w2w = torch.view_as_complex(w1h.float()), torch.view_as_complex(w2h.float())
    y_ref, gx_ref, gw1_ref, gw2_ref = _reference(xb.float(), w1w, w2w, gyb.float())
    xd = xb.to(dev()).requires_grad_(True)
    w1d, w2d = w1h.to(dev()).requires_grad_(True), w2h.to(dev()).requires_grad_(True)
    y = spectral_conv2d_mixed(xd, w1d, w2d, S, S)
    assert y.dtype == torch.bfloat16
    y.backward(gyb.to(dev()))
    assert xd.grad.dtype == torch.bfloat16 and w1d.grad.dtype == torch.float16 and w1d.grad.shape == w1d.shape
    assert rel_err(y.detach().float().cpu().numpy(), y_ref.numpy()) < TOL_BF16
    assert rel_err(xd.grad.float().cpu().numpy(), gx_ref.numpy()) < TOL_BF16
    # weight gradients are accumulated in f32 / c64 and rounded once to the storage format (fp16: 2^-11 relative)
    assert rel_err(torch.view_as_complex(w1d.grad.float().cpu()).numpy(), gw1_ref.numpy()) < 1e-3
    assert rel_err(torch.view_as_complex(w2d.grad.float().cpu()).numpy(), gw2_ref.numpy()) < 1e-3


def test_c5_model_training_step_float32():
    """UNO_9(3, 64, pad=5) at S = 1024 (padded 1089 x 1089), batch 2: two training steps - finite loss that decreases under Adam,
    bit-identical repeat from the same state (determinism of every kernel on the path), gradients of every parameter non-zero."""
    from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
    def run():
        torch.manual_seed(0)
        model = UNO_9(3, 64, pad=5).to(dev())
        tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
        a, u = synthetic_darcy_batch(2, S, 77, dev())
        l0 = float(tr.step(a, u))
        gn = {k: float(torch.linalg.vector_norm(p.grad)) for k, p in model.named_parameters()}
        l1 = float(tr.step(a, u))
        return l0, l1, gn
    l0, l1, gn = run()
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0
    zero_ok = {"conv1.w.conv.bias", "conv4.w.conv.bias"}        # biases in front of an InstanceNorm: exactly-zero true gradient
    assert all(v > 0 for k, v in gn.items() if k not in zero_ok), [k for k, v in gn.items() if v == 0]
    l0b, l1b, _ = run()
    assert (l0, l1) == (l0b, l1b)


def test_c5_model_training_step_mixed_precision():
    """The configs[4] workload itself: UNO_9(3, 64, pad=5) at S = 1024 with bf16 activations + fp16 spectral weights (f32
    accumulation, f32 master weights / Adam), batch 2: finite decreasing loss that tracks the float32 step from the same init
    (2 %), deterministic repeat."""
    from uno_amd.harness import DarcyTrainer, MixedDarcyTrainer, UNO_9, synthetic_darcy_batch
    a, u = synthetic_darcy_batch(2, S, 77, dev())
    def run(cls):
        torch.manual_seed(0)
        model = UNO_9(3, 64, pad=5).to(dev())
        tr = cls(model, lr=1e-3, weight_decay=1e-3)
        return [float(tr.step(a, u)) for _ in range(2)]
    lm, lf = run(MixedDarcyTrainer), run(DarcyTrainer)
    assert all(np.isfinite(lm)) and lm[1] < lm[0]
    assert np.allclose(lm, lf, rtol=2e-2)
    assert run(MixedDarcyTrainer) == lm
