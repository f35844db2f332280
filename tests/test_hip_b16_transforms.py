"""K1-B / K3-B (csrc/dft2d_b16.hip): the pruned transforms of bfloat16 images with their row stage on the bf16 MFMA, through the C ABI,
against the float64 dense-DFT oracle (reference integral_operators.py:187, 206: rfft2 / irfft2 of the truncated spectrum).  pytest -m gpu

Tolerances: the forward transform's output is a complex64 spectrum - the only approximation is the twiddle operand split into
hi + lo bf16 (relative error 2^-17 per factor), asserted at TOL = 2e-5 like the f32 kernels (measured ~4e-6); the inverse writes
bf16 images: TOL_BF16 = 3e-3 (half an ulp of bf16 is 2^-9 relative per element), and before the final rounding its three-product
row stage is within 2e-5 of the f32 kernel - asserted by comparing with the f32 kernel's output rounded to bf16 ulp-wise."""
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


# n_img, H, W, m1, m2: one chunk / several chunks / ragged last k-step / odd widths (rows 2-byte aligned) / ragged last row tile /
# one..eight waves per image (n_img against 256 CUs) / every (NT, MT) class the kernels are compiled for
SHAPES = [
    (3, 16, 64, 4, 5), (2, 21, 66, 4, 5), (2, 40, 130, 17, 20), (1, 85, 85, 12, 12), (2, 111, 111, 8, 8), (1, 421, 421, 20, 20),
    (2, 64, 256, 8, 16), (2, 64, 257, 8, 17), (1, 33, 1089, 16, 32), (3, 272, 272, 8, 8), (2, 100, 544, 18, 18), (1, 128, 1024, 32, 32),
    (600, 16, 67, 6, 6), (1100, 24, 72, 5, 9), (2100, 17, 65, 3, 3), (5, 48, 300, 40, 16), (4, 50, 96, 20, 33), (2, 19, 700, 8, 48),
    (2, 32, 257, 8, 16), (1, 16, 259, 4, 4), (3, 64, 289, 8, 8), (1, 48, 513, 6, 20), (2, 31, 1025, 5, 9),      # odd widths on even heights: the tensor's last pixel
]


def _b16_path(H, W, m1, m2):
    """mirror of dft2d_b16_applies (csrc/dft2d_b16.hip): the shapes the bf16-MFMA kernels take; the others keep the f32-MFMA forms"""
    NT, MT = (m2 + 15) // 16, (2 * m1 + 15) // 16
    return W >= 64 and H >= 16 and m1 <= 40 and m2 <= 32 and NT * MT <= 8


def _images(n, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 1, H, W, generator=g).bfloat16()


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_transform_of_bf16_images(shape):
    from uno_amd import _native
    n, H, W, m1, m2 = shape
    x = _images(n, H, W, 31 * H + W)
    _native.profile_begin(8)
    got = _native.dft2d_forward(x.to(dev()), m1, m2, scale=0.5, hermitian_cols=True, mask_overlap=True)
    names = [r[0] for r in _native.profile_end()]
    assert any("dft2d_fwd_b16_kernel" in k for k in names) == _b16_path(H, W, m1, m2), names
    sub = slice(0, n, max(1, n // 7))                          # the oracle on a spread of the images
    want = so.truncated_rfft2_dense(x[sub].double().numpy(), m1, m2) * (H * W) * 0.5             # oracle normalises by 1 / (H W)
    want = want * so.hermitian_weights(W, m2)[None, None, None, :] * so.later_wins_mask(H, m1)[None, None, :, None]
    assert rel_err(got[sub].cpu().numpy(), want) < TOL


@pytest.mark.parametrize("shape", SHAPES)
def test_inverse_transform_to_bf16_images(shape):
    from uno_amd import _native
    n, H, W, m1, m2 = shape
    g = torch.Generator().manual_seed(17 * H + W)
    O = torch.randn(n, 1, 2 * m1, m2, dtype=torch.cfloat, generator=g)
    _native.profile_begin(8)
    got = _native.dft2d_inverse(O.to(dev()), H, W, scale=0.25, dtype=torch.bfloat16)
    names = [r[0] for r in _native.profile_end()]
    assert any("dft2d_inv_b16_kernel" in k for k in names) == _b16_path(H, W, m1, m2), names
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (n, 1, H, W)
    sub = slice(0, n, max(1, n // 7))
    want = so.truncated_irfft2_dense(O[sub].numpy().astype(np.complex128), H, W, m1, m2) * 0.25
    assert rel_err(got[sub].float().cpu().numpy(), want) < TOL_BF16
    # before its final rounding the result is as good as the f32 kernel's: the two agree to one bf16 ulp almost everywhere
    f32 = _native.dft2d_inverse(O.to(dev()), H, W, scale=0.25)
    ulp = torch.maximum(f32.abs(), torch.tensor(1e-30, device=f32.device)).log2().floor().exp2() * 2.0 ** -7
    off = ((got.float() - f32).abs() > 0.5001 * ulp + 1e-6 * f32.abs().max()).float().mean()
    assert float(off) < 2e-3, float(off)


def test_grouped_spectrum_layout_two_sources():
    """K1-B writes the spectra of a (B, C1) batch into channels [off, off + C1) of a (B, Ctot) spectrum tensor and K3-B reads a channel
    range of one (the two-source operator block, uno_dft2d_*_grouped)."""
    from uno_amd import _native
    B, C1, C2, H, W, m1, m2 = 3, 4, 2, 40, 96, 6, 7
    g = torch.Generator().manual_seed(5)
    x1 = torch.randn(B, C1, H, W, generator=g).bfloat16().to(dev())
    x2 = torch.randn(B, C2, H, W, generator=g).bfloat16().to(dev())
    out = torch.zeros(B, C1 + C2, 2 * m1, m2, dtype=torch.cfloat, device=dev())
    _native.dft2d_forward(x1, m1, m2, 1.0 / (H * W), out=out, channel_offset=0)
    _native.dft2d_forward(x2, m1, m2, 1.0 / (H * W), out=out, channel_offset=C1)
    want = so.truncated_rfft2_dense(torch.cat([x1, x2], 1).float().cpu().double().numpy(), m1, m2)
    assert rel_err(out.cpu().numpy(), want) < TOL
    back = _native.dft2d_inverse(out, H, W, 1.0, False, False, channels=C2, channel_offset=C1, dtype=torch.bfloat16)
    full = _native.dft2d_inverse(out, H, W, 1.0, False, False)
    assert rel_err(back.float().cpu().numpy(), full[:, C1:].cpu().numpy()) < TOL_BF16


def test_transforms_are_deterministic_and_leave_neighbouring_memory_alone():
    """Same bits on repeat; images / spectra embedded in larger poisoned buffers keep their margins (ragged rows: 2-byte aligned 16-byte
    stores, partial last chunk)."""
    from uno_amd import _native
    n, H, W, m1, m2 = 3, 37, 333, 9, 11
    g = torch.Generator().manual_seed(1)
    pad = 4096
    xbuf = torch.full((n * H * W + 2 * pad,), 7.0).bfloat16().to(dev())
    x = xbuf[pad:pad + n * H * W].view(n, 1, H, W)
    x.copy_(torch.randn(n, 1, H, W, generator=g).bfloat16())
    a = _native.dft2d_forward(x, m1, m2)
    b = _native.dft2d_forward(x, m1, m2)
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))
    y1 = _native.dft2d_inverse(a, H, W, dtype=torch.bfloat16)
    y2 = _native.dft2d_inverse(a, H, W, dtype=torch.bfloat16)
    assert torch.equal(y1, y2)
    # inverse into the middle of a poisoned buffer through the raw entry point
    ybuf = torch.full((n * H * W + 2 * pad,), 3.0).bfloat16().to(dev())
    with torch.cuda.device(dev()):
        rc = _native.lib().uno_dft2d_inverse_bf16(_native._ptr(a), ybuf.data_ptr() + 2 * pad, n, H, W, m1, m2, 1.0, 1, 1, _native._stream(a))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(ybuf[pad:pad + n * H * W].view(n, 1, H, W), y1)
    three = torch.tensor(3.0).bfloat16().to(dev())
    assert bool((ybuf[:pad] == three).all()) and bool((ybuf[pad + n * H * W:] == three).all())
