"""Parity of the HIP spectral-convolution path (through the C ABI) against the oracle and the
reference-generated golden vectors.  Needs a real MI355X:  pytest -m gpu

Tolerances (float32 path; stated as relative L2 error  ||a-b|| / ||b||):
  TOL      = 2e-5   full operator and every stage vs the float64 dense oracle / golden vectors.
             The HIP path accumulates direct DFT sums of length <= ~1100 in exact-f32 MFMA, the
             reference uses float32 FFTs; both sit at a few 1e-7..1e-6 of the float64 result.
"""
import numpy as np
import pytest
import torch

from conftest import Case, load_cases, rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
TOL = 2e-5

Z2, NAMES2 = load_cases("spectral2d.npz")
CASES2 = [n for n in NAMES2 if n not in ("fp64_in",)]


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


# ------------------------------------------------------------------ stage level: pruned DFTs
DFT_SHAPES = [
    # n_img, H, W, m1, m2
    (3, 16, 16, 4, 5), (2, 21, 18, 4, 5), (2, 23, 23, 11, 12), (1, 40, 44, 17, 20), (2, 85, 85, 12, 12),
    (1, 90, 90, 18, 18), (2, 111, 111, 8, 8), (1, 223, 223, 8, 8), (1, 64, 66, 32, 33), (2, 10, 14, 7, 5),
    (1, 7, 130, 3, 40), (1, 130, 6, 40, 4), (1, 1, 2, 1, 2), (1, 421, 421, 20, 20),
    # many small images: the plane-batched kernels K1p / K3p (n_img >= 128, H*W <= 2048, W <= 64, 2 m1 <= 48, 2 m2 <= 32)
    (130, 16, 16, 6, 6), (200, 64, 20, 16, 8), (129, 64, 13, 22, 5), (128, 48, 26, 14, 8), (160, 32, 32, 14, 14),
    (128, 21, 18, 4, 5), (131, 23, 23, 11, 12), (128, 16, 15, 8, 8), (1100, 10, 14, 5, 5), (150, 4, 4, 2, 3),
    (128, 5, 7, 2, 4), (128, 33, 61, 16, 16), (128, 32, 64, 16, 16), (4200, 64, 26, 22, 8), (128, 50, 40, 24, 3),
]


@pytest.mark.parametrize("shape", DFT_SHAPES)
@pytest.mark.parametrize("flags", [(False, False), (True, True)])
def test_dft2d_forward_stage(shape, flags):
    from uno_amd import _native
    n, H, W, m1, m2 = shape
    herm, mask = flags
    rng = np.random.default_rng(1000 + H * 7 + W)
    x = rng.standard_normal((n, 1, H, W)).astype(np.float32)
    got = _native.dft2d_forward(cu(x), m1, m2, scale=0.5, hermitian_cols=herm, mask_overlap=mask).cpu().numpy()
    ref = so.truncated_rfft2_dense(x, m1, m2) * (H * W) * 0.5
    if herm:
        ref = ref * so.hermitian_weights(W, m2)[None, None, None, :]
    if mask:
        ref = ref * so.later_wins_mask(H, m1)[None, None, :, None]
    assert got.shape == ref.shape
    assert rel_err(got, ref) < TOL
    if mask:
        dead = so.later_wins_mask(H, m1) == 0
        assert np.all(got[:, :, dead, :] == 0)


@pytest.mark.parametrize("shape", DFT_SHAPES)
@pytest.mark.parametrize("flags", [(False, False), (True, True)])
def test_dft2d_inverse_stage(shape, flags):
    from uno_amd import _native
    n, H, W, m1, m2 = shape
    herm, mask = flags
    rng = np.random.default_rng(2000 + H * 7 + W)
    O = (rng.standard_normal((n, 1, 2 * m1, m2)) + 1j * rng.standard_normal((n, 1, 2 * m1, m2))).astype(np.complex64)
    got = _native.dft2d_inverse(cu(O), H, W, scale=0.25, hermitian_cols=herm, mask_overlap=mask).cpu().numpy()
    Gh = so._dft(so.corner_rows(H, m1), H, +1.0)
    Gw = so._dft(np.arange(m2), W, +1.0)
    keep = so.later_wins_mask(H, m1) if mask else np.ones(2 * m1)
    c = so.hermitian_weights(W, m2) if herm else np.ones(m2)
    U = np.einsum("bojl,jh->bohl", O.astype(np.complex128) * keep[None, None, :, None], Gh)
    ref = 0.25 * np.einsum("bohl,lw->bohw", U * c, Gw).real
    assert rel_err(got, ref) < TOL


def test_plane_kernels_grouped_spectrum_layout():
    """Two-source blocks at a coarse level: the plane-batched kernels honour the (group, stride, offset) spectrum layout."""
    from uno_amd import _native
    B, C1, C2, H, W, m1, m2 = 8, 24, 16, 16, 20, 6, 7
    rng = np.random.default_rng(77)
    x1 = rng.standard_normal((B, C1, H, W)).astype(np.float32)
    x2 = rng.standard_normal((B, C2, H, W)).astype(np.float32)
    spec = torch.zeros((B, C1 + C2, 2 * m1, m2), dtype=torch.complex64, device=dev())
    _native.dft2d_forward(cu(x1), m1, m2, out=spec, channel_offset=0)
    _native.dft2d_forward(cu(x2), m1, m2, out=spec, channel_offset=C1)
    ref = so.truncated_rfft2_dense(np.concatenate([x1, x2], axis=1), m1, m2) * (H * W)
    assert rel_err(spec.cpu().numpy(), ref) < TOL
    whole = _native.dft2d_inverse(spec, H, W, scale=1.0).cpu().numpy()
    part1 = _native.dft2d_inverse(spec, H, W, scale=1.0, channels=C1, channel_offset=0).cpu().numpy()
    part2 = _native.dft2d_inverse(spec, H, W, scale=1.0, channels=C2, channel_offset=C1).cpu().numpy()
    assert np.array_equal(part1, whole[:, :C1]) and np.array_equal(part2, whole[:, C1:])


# ------------------------------------------------------------------ stage level: per-mode GEMMs
MIX_SHAPES = [
    # B, Ci, Co, ncorner, modes-per-corner shape
    (2, 3, 4, 2, (4, 5)), (16, 64, 64, 2, (20, 20)), (8, 32, 32, 2, (12, 12)), (5, 7, 9, 2, (3, 11)),
    (17, 20, 33, 2, (2, 9)), (1, 1, 1, 2, (1, 1)), (4, 6, 5, 4, (3, 3, 2)), (3, 48, 24, 2, (18, 18)),
    # long channel loop on a small grid: the 8-mode workgroup variant (forward K = Ci, input gradient K = Co)
    (4, 96, 40, 2, (3, 3)), (3, 20, 128, 2, (5, 4)), (2, 130, 100, 4, (3, 2, 2)), (16, 256, 256, 2, (8, 8)), (5, 192, 48, 2, (3, 7)),
    # mode counts with little padding to a multiple of 16: the 4x4x1 form (16 modes = the 16 blocks of an instruction) with ragged
    # rows / columns / reduction lengths, its K split (few tasks, long K), the 16 x 16-per-wave weight-gradient form (K = B <= 32)
    (5, 7, 9, 2, (9, 11)), (17, 20, 33, 2, (7, 9)), (3, 130, 100, 4, (6, 5, 3)), (8, 32, 64, 4, (6, 6, 5)), (32, 48, 96, 2, (14, 14)),
    (9, 70, 18, 2, (10, 11)), (2, 64, 8, 2, (16, 16)),
    # odd channel counts >= 96 on few modes (found by the random-shape test: the 8-mode variant's tail load assumed even K)
    (17, 115, 45, 2, (3, 4)), (4, 97, 40, 2, (3, 6)), (3, 40, 101, 2, (2, 5)),
]


@pytest.mark.parametrize("shape", MIX_SHAPES)
def test_mode_gemms(shape):
    from uno_amd import _native
    B, Ci, Co, nc, mshape = shape
    rng = np.random.default_rng(3000 + B + Ci * 3 + Co * 5)
    cplx = lambda *s: (rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(np.complex64)
    X = cplx(B, Ci, nc, *mshape)
    gO = cplx(B, Co, nc, *mshape)
    ws = [cplx(Ci, Co, *mshape) for _ in range(nc)]
    wd = [cu(w) for w in ws]
    O = _native.mode_mix(cu(X), wd, 0).cpu().numpy()
    gX = _native.mode_mix(cu(gO), wd, 1).cpu().numpy()
    gW = [g.cpu().numpy() for g in _native.mode_wgrad(cu(X), cu(gO), ws[0].shape, nc)]
    X64, gO64 = X.astype(np.complex128), gO.astype(np.complex128)
    for c in range(nc):
        w64 = ws[c].astype(np.complex128)
        assert rel_err(O[:, :, c], np.einsum("bi...,io...->bo...", X64[:, :, c], w64)) < TOL
        assert rel_err(gX[:, :, c], np.einsum("bo...,io...->bi...", gO64[:, :, c], np.conj(w64))) < TOL
        assert rel_err(gW[c], np.einsum("bi...,bo...->io...", np.conj(X64[:, :, c]), gO64[:, :, c])) < TOL


def _random_mix_shapes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nc = int(rng.choice([2, 4]))
        mshape = tuple(int(v) for v in rng.integers(1, 13, size=2 if nc == 2 else 3))
        out.append((int(rng.integers(1, 34)), int(rng.integers(1, 140)), int(rng.integers(1, 140)), nc, mshape))
    return out


@pytest.mark.parametrize("shape", _random_mix_shapes(24, 777), ids=lambda s: f"B{s[0]}-{s[1]}x{s[2]}-c{s[3]}-m" + "x".join(map(str, s[4])))
def test_mode_gemms_random_shapes(shape):
    """K2 in all three roles on seeded random (batch, channels, mode-count) combinations: both kernel families (4x4x1 blocks and
    the LDS-staged form for badly padded mode counts), every tile configuration and K split."""
    test_mode_gemms(shape)


# ------------------------------------------------------------------ full operator vs golden vectors
@pytest.mark.parametrize("name", CASES2)
def test_golden_forward_backward(name):
    from uno_amd.integral_operators import spectral_conv2d
    c = Case(Z2, name)
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = [int(v) for v in c.meta]
    x = cu(c.x).requires_grad_(True)
    w1 = cu(c.w1).requires_grad_(True)
    w2 = cu(c.w2).requires_grad_(True)
    y = spectral_conv2d(x, w1, w2, Ho, Wo)
    assert y.dtype == torch.float32 and tuple(y.shape) == c.y.shape
    assert rel_err(y.detach().cpu().numpy(), c.y) < TOL
    y.backward(cu(c.gy))
    assert rel_err(x.grad.cpu().numpy(), c.gx) < TOL
    assert rel_err(w1.grad.cpu().numpy(), c.gw1) < TOL
    assert rel_err(w2.grad.cpu().numpy(), c.gw2) < TOL
    # later-wins: weights1 rows overwritten by weights2's corner receive exactly zero gradient
    zero_ref = c.gw1 == 0
    if zero_ref.any():
        assert np.all(w1.grad.cpu().numpy()[zero_ref] == 0)


def test_noncontiguous_input_matches():
    from uno_amd.integral_operators import spectral_conv2d
    c = Case(Z2, "noncontig")
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = [int(v) for v in c.meta]
    xt = cu(np.ascontiguousarray(np.swapaxes(c.x, -1, -2))).transpose(-1, -2)
    assert not xt.is_contiguous()
    y = spectral_conv2d(xt, cu(c.w1), cu(c.w2), Ho, Wo)
    assert rel_err(y.cpu().numpy(), c.y) < TOL


# ------------------------------------------------------------------ seeded cases vs the dense oracle
SEEDED = [
    # B, Ci, Co, H, W, Ho, Wo, m1, m2
    (8, 32, 32, 85, 85, 85, 85, 12, 12),        # BASELINE config 1 block
    (2, 8, 16, 90, 90, 45, 45, 18, 18),         # UNO_9.conv0 geometry
    (2, 16, 8, 45, 45, 90, 90, 18, 18),         # UNO_9.conv5 geometry
    (2, 4, 4, 64, 64, 64, 64, 32, 33),          # Nyquist column + full rows
    (3, 5, 3, 37, 50, 29, 31, 9, 13),
    (1, 2, 3, 96, 100, 88, 96, 40, 48),         # the largest compiled mode counts (modes1 = 40, modes2 = 48)
    (1, 2, 2, 128, 136, 128, 136, 32, 32),      # BASELINE config 5 modes (32, 32) on a small grid
]


@pytest.mark.parametrize("cfg", SEEDED)
def test_seeded_vs_dense_oracle(cfg):
    from uno_amd.integral_operators import spectral_conv2d
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = cfg
    rng = np.random.default_rng(sum(cfg))
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    sc = (1 / (2 * Ci)) ** 0.5
    w1 = (sc * (rng.standard_normal((Ci, Co, m1, m2)) + 1j * rng.standard_normal((Ci, Co, m1, m2)))).astype(np.complex64)
    w2 = (sc * (rng.standard_normal((Ci, Co, m1, m2)) + 1j * rng.standard_normal((Ci, Co, m1, m2)))).astype(np.complex64)
    gy = rng.standard_normal((B, Co, Ho, Wo)).astype(np.float32)
    y_ref, X = so.spectral_conv2d_dense(x, w1, w2, Ho, Wo)
    gx_ref, gw1_ref, gw2_ref, _, _ = so.spectral_conv2d_dense_bwd(gy, X, w1, w2, H, W)
    xd, w1d, w2d = cu(x).requires_grad_(True), cu(w1).requires_grad_(True), cu(w2).requires_grad_(True)
    y = spectral_conv2d(xd, w1d, w2d, Ho, Wo)
    y.backward(cu(gy))
    assert rel_err(y.detach().cpu().numpy(), y_ref) < TOL
    assert rel_err(xd.grad.cpu().numpy(), gx_ref) < TOL
    assert rel_err(w1d.grad.cpu().numpy(), gw1_ref) < TOL
    assert rel_err(w2d.grad.cpu().numpy(), gw2_ref) < TOL


# ------------------------------------------------------------------ BASELINE config 2 at full size
def test_full_size_421_vs_fft_oracle_and_properties():
    """Darcy 421x421, 64 ch, modes 20, batch 16 (BASELINE.json configs[1], block level): direct
    comparison with the FFT-sequence oracle on the host plus size-independent properties
    (linearity, adjoint identity <A x, g> = <x, A^T g>, determinism)."""
    from uno_amd.integral_operators import spectral_conv2d
    B, C, S, m = 16, 64, 421, 20
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, C, S, S, generator=g)
    sc = (1 / (2 * C)) ** 0.5
    w1 = sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)
    w2 = sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)
    gy = torch.randn(B, C, S, S, generator=g)

    xr, w1r, w2r = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    y_ref = so.spectral_conv2d_fft(xr, w1r, w2r, S, S)
    y_ref.backward(gy)

    xd, w1d, w2d = x.to(dev()).requires_grad_(True), w1.to(dev()).requires_grad_(True), w2.to(dev()).requires_grad_(True)
    y = spectral_conv2d(xd, w1d, w2d, S, S)
    y.backward(gy.to(dev()))
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < TOL
    assert rel_err(xd.grad.cpu().numpy(), xr.grad.numpy()) < TOL
    assert rel_err(w1d.grad.cpu().numpy(), w1r.grad.numpy()) < TOL
    assert rel_err(w2d.grad.cpu().numpy(), w2r.grad.numpy()) < TOL

    with torch.no_grad():
        # determinism: identical bits on a second run
        y2 = spectral_conv2d(xd, w1d, w2d, S, S)
        assert torch.equal(y2, y.detach())
        # linearity in x
        x2 = torch.randn(B, C, S, S, generator=g).to(dev())
        lhs = spectral_conv2d(2.0 * xd - 3.0 * x2, w1d, w2d, S, S)
        rhs = 2.0 * y.detach() - 3.0 * spectral_conv2d(x2, w1d, w2d, S, S)
        assert rel_err(lhs.cpu().numpy(), rhs.cpu().numpy()) < TOL
        # adjoint identity in float64 accumulation
        a = torch.dot(y.detach().double().flatten(), gy.to(dev()).double().flatten())
        b = torch.dot(xd.detach().double().flatten(), xd.grad.double().flatten())
        assert abs(a.item() - b.item()) <= 1e-5 * max(abs(a.item()), abs(b.item()))


# ------------------------------------------------------------------ interface behaviour
def test_module_interface_and_errors():
    from uno_amd.integral_operators import SpectralConv2d_Uno
    torch.manual_seed(0)
    conv = SpectralConv2d_Uno(3, 4, 16, 16, 4, 5)
    x_cpu = torch.randn(2, 3, 20, 20)
    with pytest.raises(RuntimeError):           # no CPU fallback on the product path
        conv(x_cpu)
    conv = conv.to(dev())
    x = x_cpu.to(dev())
    y = conv(x)
    assert tuple(y.shape) == (2, 4, 16, 16) and y.dtype == torch.float32
    y2 = conv(x, 12, 10)
    assert tuple(y2.shape) == (2, 4, 12, 10)
    assert (conv.dim1, conv.dim2) == (12, 10)   # dims override persists (reference :182-184)
    assert tuple(conv(x).shape) == (2, 4, 12, 10)
    with pytest.raises(RuntimeError):           # float64 input: the reference raises too
        conv(x.double())
    with pytest.raises(RuntimeError):           # modes2 > W//2+1 of the requested output
        conv(x, 12, 6)
    with pytest.raises(RuntimeError):
        conv(torch.randn(2, 5, 20, 20, device=dev()))


def test_empty_batch():
    from uno_amd.integral_operators import spectral_conv2d
    w = torch.randn(3, 4, 2, 3, dtype=torch.cfloat, device=dev())
    x = torch.zeros(0, 3, 8, 8, device=dev())
    assert tuple(spectral_conv2d(x, w, w, 8, 8).shape) == (0, 4, 8, 8)


@pytest.mark.parametrize("cfg", [(3, 4, 5, 37, 50, 9), (2, 6, 3, 64, 32, 17), (1, 1, 2, 9, 9, 5), (4, 8, 8, 421, 211, 20)])
def test_spectral_conv1d_runs_on_the_2d_kernels(cfg):
    """SpectralConv1d_Uno (reference integral_operators.py:7-72) on the GPU = the 2-D layer on a one-row grid; compared with
    the module's stock torch.fft path on the CPU (forward, input gradient, weight gradient)."""
    from uno_amd.integral_operators import SpectralConv1d_Uno
    B, Ci, Co, N, dim1, modes = cfg
    torch.manual_seed(N)
    ref = SpectralConv1d_Uno(Ci, Co, dim1, modes)
    gpu = SpectralConv1d_Uno(Ci, Co, dim1, modes).to(dev())
    gpu.load_state_dict(ref.state_dict())
    x = torch.randn(B, Ci, N)
    gy = torch.randn(B, Co, dim1)
    xr, xg = x.clone().requires_grad_(True), x.to(dev()).requires_grad_(True)
    yr = ref(xr); yr.backward(gy)
    yg = gpu(xg); yg.backward(gy.to(dev()))
    assert yg.shape == yr.shape
    assert rel_err(yg.detach().cpu().numpy(), yr.detach().numpy()) < TOL
    assert rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) < TOL
    assert rel_err(gpu.weights1.grad.cpu().numpy(), ref.weights1.grad.numpy()) < TOL


# ------------------------------------------------------------------ mode counts beyond the MFMA kernels' compiled range
@pytest.mark.parametrize("cfg", [
    # B, Ci, Co, H, W, Ho, Wo, m1, m2
    (2, 3, 4, 100, 100, 100, 100, 49, 50),      # the reference's DEFAULT modes: dim1//2 - 1, dim2//2 (integral_operators.py:153-158)
    (1, 2, 2, 90, 120, 96, 110, 44, 20),        # only modes1 beyond 40
    (1, 2, 3, 64, 130, 64, 140, 8, 60),         # only modes2 beyond 48
    (1, 1, 2, 50, 50, 42, 44, 42, 10),          # overlap 2*m1 > H' with the any-mode form (later-wins mask)
])
def test_any_mode_count_vs_dense_oracle(cfg):
    from uno_amd.integral_operators import spectral_conv2d
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Ci, H, W, generator=g)
    w1 = 0.2 * torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g)
    w2 = 0.2 * torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g)
    gy = torch.randn(B, Co, Ho, Wo, generator=g)
    xd, w1d, w2d = (t.to(dev()).requires_grad_(True) for t in (x, w1, w2))
    y = spectral_conv2d(xd, w1d, w2d, Ho, Wo)
    y.backward(gy.to(dev()))
    y_ref, X = so.spectral_conv2d_dense(x.numpy(), w1.numpy(), w2.numpy(), Ho, Wo)
    gx_ref, gw1_ref, gw2_ref, _, _ = so.spectral_conv2d_dense_bwd(gy.numpy(), X, w1.numpy(), w2.numpy(), H, W)
    assert rel_err(y.detach().cpu().numpy(), y_ref) < TOL
    assert rel_err(xd.grad.cpu().numpy(), gx_ref) < TOL
    assert rel_err(w1d.grad.cpu().numpy(), gw1_ref) < TOL
    assert rel_err(w2d.grad.cpu().numpy(), gw2_ref) < TOL


def test_default_modes_module_runs_like_the_reference():
    """SpectralConv2d_Uno built WITHOUT modes (reference defaults dim1//2 - 1, dim2//2) against the reference's op sequence."""
    from uno_amd.integral_operators import SpectralConv2d_Uno
    torch.manual_seed(11)
    conv = SpectralConv2d_Uno(3, 2, 120, 120)
    assert (conv.modes1, conv.modes2) == (59, 60)
    x = torch.randn(2, 3, 120, 120)
    y_ref = so.spectral_conv2d_fft(x, conv.weights1.detach(), conv.weights2.detach(), 120, 120)
    y = conv.to(dev())(x.to(dev()))
    assert rel_err(y.detach().cpu().numpy(), y_ref.numpy()) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("B,Ci,Co,m1,m2,acc", [(16, 256, 64, 18, 18, False), (16, 64, 128, 18, 18, True), (8, 32, 48, 6, 7, False), (3, 20, 12, 5, 4, True),
                                               (16, 128, 256, 8, 8, False)])
def test_paired_backward_gemms_equal_the_two_single_launches(B, Ci, Co, m1, m2, acc):
    """uno_mode_backward (both per-mode GEMMs of a backward pass from one launch where the kernels allow, the autograd adjoints of the
    einsum at reference integral_operators.py:178-179) against uno_mode_mix(op 1) + uno_mode_wgrad and against the float64 einsums."""
    from uno_amd import _native
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + Ci + Co + m1)
    xt = torch.randn(B, Ci, 2, m1 * m2, dtype=torch.complex64, generator=g).to(dev)
    go = torch.randn(B, Co, 2, m1 * m2, dtype=torch.complex64, generator=g).to(dev)
    ws = [torch.randn(Ci, Co, m1, m2, dtype=torch.complex64, generator=g).to(dev) for _ in range(2)]
    base = [torch.randn(Ci, Co, m1, m2, dtype=torch.complex64, generator=g).to(dev) for _ in range(2)]
    out = [b.clone() for b in base] if acc else None
    gX, gws = _native.mode_backward(xt, go, ws, out=out, accumulate=acc)
    gX1 = _native.mode_mix(go, ws, 1)
    gw1 = _native.mode_wgrad(xt, go, tuple(ws[0].shape), 2)
    assert torch.equal(gX.view_as(gX1), gX1)                       # the same kernel bodies: bit for bit
    W = torch.stack([w.reshape(Ci, Co, -1) for w in ws], 2).to(torch.complex128)          # (Ci, Co, 2, M)
    ref_gx = torch.einsum("bocm,iocm->bicm", go.to(torch.complex128), W.conj())
    ref_gw = torch.einsum("bicm,bocm->iocm", xt.to(torch.complex128).conj(), go.to(torch.complex128))
    assert float((gX.view(B, Ci, 2, -1).to(torch.complex128) - ref_gx).abs().max()) < 2e-5 * float(ref_gx.abs().max())
    for c in range(2):
        want = ref_gw[:, :, c].reshape(Ci, Co, m1, m2) + (base[c].to(torch.complex128) if acc else 0)
        assert float((gws[c].to(torch.complex128) - want).abs().max()) < 2e-5 * float(want.abs().max())
        if not acc:
            assert torch.equal(gws[c], gw1[c])
