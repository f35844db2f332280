"""Red-zone test ("race detection / sanitizers" row): every output, workspace and gradient buffer the product
allocates from Python is embedded in a larger POISONED allocation, and after each workload the margins must be bit-unchanged - an
out-of-bounds store of any kernel (clamped raw-buffer addressing, ragged tails, 2-byte-aligned 16-byte stores, split-K workspaces)
fails the test instead of silently corrupting a neighbouring tensor.  The parity tests compare values only.  pytest -m gpu

How: inside the fixture torch.empty / empty_like / zeros / zeros_like (the only allocation calls of uno_amd/_native.py,
uno_amd/integral_operators.py and uno_amd/harness/*) return views into [64 KiB guard | tensor | 64 KiB guard] blocks filled with 0xA5;
the workloads are the library's own ragged-shape cases (odd grids, prime sizes, overlapping corners, partial tiles, one- and
two-source blocks, bf16 forms, 3-D volumes and planes, the roll-out's stacked spectra) driven through the public modules, so every C
entry point that the models reach runs with guarded outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GUARD = 65536          # 64 KiB on both sides (round 4 used 4 KiB: a far stride - a row or plane pitch off by one - jumped over it)
PATTERN = 0xA5


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class RedZone:
    def __init__(self, monkeypatch):
        self.records = []
        self._empty, self._zeros = torch.empty, torch.zeros
        monkeypatch.setattr(torch, "empty", self.empty)
        monkeypatch.setattr(torch, "zeros", self.zeros)
        monkeypatch.setattr(torch, "empty_like", self.empty_like)
        monkeypatch.setattr(torch, "zeros_like", self.zeros_like)

    def _shape(self, size):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            return tuple(int(v) for v in size[0])
        return tuple(int(v) for v in size)

    def _guarded(self, shape, dtype, device):
        dtype = dtype or torch.get_default_dtype()
        item = self._empty((), dtype=dtype).element_size()
        n = int(np.prod(shape)) if len(shape) else 1
        nbytes = n * item
        raw = self._empty(GUARD + nbytes + GUARD, dtype=torch.uint8, device=device)
        raw.fill_(PATTERN)
        self.records.append((raw, nbytes))
        return raw[GUARD:GUARD + nbytes].view(dtype).view(shape)

    def empty(self, *size, dtype=None, device=None, **kw):
        d = torch.device(device) if device is not None else None
        if d is None or d.type != "cuda" or kw:
            return self._empty(*size, dtype=dtype, device=device, **kw)
        return self._guarded(self._shape(size), dtype, d)

    def zeros(self, *size, dtype=None, device=None, **kw):
        d = torch.device(device) if device is not None else None
        if d is None or d.type != "cuda" or kw:
            return self._zeros(*size, dtype=dtype, device=device, **kw)
        return self._guarded(self._shape(size), dtype, d).zero_()

    def empty_like(self, t, **kw):
        if not t.is_cuda or kw:
            return self._empty(t.shape, dtype=kw.pop("dtype", t.dtype), device=kw.pop("device", t.device), **kw)
        return self._guarded(tuple(t.shape), t.dtype, t.device)

    def zeros_like(self, t, **kw):
        if not t.is_cuda or kw:
            return self._zeros(t.shape, dtype=kw.pop("dtype", t.dtype), device=kw.pop("device", t.device), **kw)
        return self._guarded(tuple(t.shape), t.dtype, t.device).zero_()

    def check(self, what):
        torch.cuda.synchronize()
        assert len(self.records) > 0, "nothing was allocated under the red-zone fixture"
        for i, (raw, nbytes) in enumerate(self.records):
            front, back = raw[:GUARD], raw[GUARD + nbytes:]
            ok = bool((front == PATTERN).all()) and bool((back == PATTERN).all())
            if not ok:
                bf = (front != PATTERN).nonzero().flatten()
                bb = (back != PATTERN).nonzero().flatten()
                raise AssertionError(f"{what}: allocation {i} ({nbytes} bytes) has a damaged guard band: "
                                     f"{bf.numel()} bytes in front (last at -{GUARD - int(bf.max()) if bf.numel() else 0}), "
                                     f"{bb.numel()} bytes behind (first at +{int(bb.min()) if bb.numel() else 0})")
        n = len(self.records)
        self.records = []
        return n


@pytest.fixture
def redzone(monkeypatch):
    return RedZone(monkeypatch)


def test_fixture_catches_a_store_past_the_end(redzone):
    t = torch.empty((3, 5), dtype=torch.float32, device=dev())
    assert t.data_ptr() % 512 == 0                      # the guarded tensor keeps the allocator's alignment (the kernels' fast paths stay reachable)
    flat = t.view(-1)
    torch.as_strided(flat, (16,), (1,)).fill_(1.0)      # one float past the end
    with pytest.raises(AssertionError, match="damaged guard band"):
        redzone.check("deliberate overrun")


SPECTRAL_2D = [  # B, Ci, Co, H, W, Ho, Wo, m1, m2: register / full-tile / half-tile / plane-batched / any-mode transforms, both K2 forms
    (2, 3, 2, 21, 18, 13, 10, 4, 5), (1, 2, 3, 45, 45, 22, 22, 8, 8), (2, 2, 2, 85, 85, 85, 85, 12, 12), (1, 3, 2, 111, 111, 223, 223, 8, 8),
    (1, 2, 2, 223, 223, 111, 111, 18, 18), (1, 2, 1, 301, 299, 150, 151, 18, 18), (1, 1, 2, 421, 421, 421, 421, 20, 20),
    (2, 2, 2, 16, 16, 16, 16, 7, 8), (3, 50, 3, 23, 23, 11, 11, 4, 4), (1, 2, 2, 20, 20, 10, 10, 8, 6), (40, 4, 4, 16, 15, 16, 15, 6, 6),
]


@pytest.mark.parametrize("cfg", SPECTRAL_2D)
def test_spectral_conv2d_forward_backward(redzone, cfg):
    from uno_amd.integral_operators import spectral_conv2d
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, H, W, generator=g).to(dev()).requires_grad_(True)
    w1 = torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g).to(dev()).requires_grad_(True)
    w2 = torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g).to(dev()).requires_grad_(True)
    y = spectral_conv2d(x, w1, w2, Ho, Wo)
    y.backward(torch.randn(B, Co, Ho, Wo, generator=g).to(dev()))
    assert redzone.check(f"spectral_conv2d {cfg}") >= 4


@pytest.mark.parametrize("cfg", [(2, 3, 2, 40, 66, 64, 70, 6, 7), (1, 2, 2, 33, 257, 33, 129, 8, 17), (2, 2, 3, 64, 130, 32, 65, 17, 20)])
def test_spectral_conv2d_mixed_precision(redzone, cfg):
    """bf16 images: the bf16-MFMA transforms (2-byte-aligned 16-byte stores of odd-length rows, partial last chunk) and the fp16-weight K2"""
    from uno_amd.integral_operators import spectral_conv2d_mixed
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = cfg
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, Ci, H, W, generator=g).bfloat16().to(dev()).requires_grad_(True)
    w1 = torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g).to(dev()).requires_grad_(True)
    w2 = torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g).to(dev()).requires_grad_(True)
    y = spectral_conv2d_mixed(x, w1, w2, Ho, Wo)
    y.backward(torch.randn(B, Co, Ho, Wo, generator=g).bfloat16().to(dev()))
    assert redzone.check(f"spectral_conv2d_mixed {cfg}") >= 4


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("geom", [(37, 37, 55, 55), (55, 55, 37, 37), (41, 41, 41, 41), (70, 66, 35, 33)])
def test_operator_block_2d(redzone, geom, normalize):
    from uno_amd.integral_operators import OperatorBlock_2D
    H, W, Ho, Wo = geom
    torch.manual_seed(3)
    blk = OperatorBlock_2D(6, 5, Ho, Wo, 5, 6, Normalize=normalize).to(dev())
    x = torch.randn(2, 6, H, W, device=dev(), requires_grad=True)
    blk(x).square().sum().backward()
    two = OperatorBlock_2D(80, 7, Ho, Wo, 5, 6, Normalize=normalize).to(dev())
    a = torch.randn(2, 64, H, W, device=dev(), requires_grad=True)
    b = torch.randn(2, 16, H, W, device=dev(), requires_grad=True)
    two.forward_cat([a, b], Ho, Wo).square().sum().backward()
    assert redzone.check(f"OperatorBlock_2D {geom} normalize={normalize}") >= 10


def test_darcy_training_steps_f32_and_mixed(redzone):
    """the whole UNO_9 step (lift, five blocks with joined skip gradients, fused projection, Adam): every kernel family of the headline
    workload at a ragged grid (S = 75: padded 80, levels 40 / 20), float32 and mixed precision"""
    from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
    from uno_amd.harness.mixed import MixedDarcyTrainer
    for cls in (DarcyTrainer, MixedDarcyTrainer):
        torch.manual_seed(0)
        model = UNO_9(3, 16, pad=5).to(dev())
        tr = cls(model, lr=1e-3, weight_decay=1e-3)
        a, u = synthetic_darcy_batch(2, 75, 5, dev())
        for _ in range(2):
            loss = tr.step(a, u)
        assert bool(torch.isfinite(loss))
        assert redzone.check(cls.__name__) >= 50


def test_ns2d_rollout_with_stacked_spectra(redzone):
    from uno_amd.harness import ComplexAdam, UNO, ns2d_rollout_loss
    torch.manual_seed(0)
    model = UNO(14, 8).to(dev())
    opt = ComplexAdam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    xx, yy = torch.randn(2, 64, 64, 10, device=dev()), torch.randn(2, 64, 64, 3, device=dev())
    for _ in range(3):                       # the third step runs on the stacks sized by the first two passes
        opt.zero_grad(set_to_none=True)
        ns2d_rollout_loss(model, xx, yy, 3).backward()
        opt.step()
    assert redzone.check("NS-2D roll-out") >= 50


def test_ns3d_training_step(redzone):
    from uno_amd.harness import ComplexAdam, Uno3D_T20, ns3d_loss
    torch.manual_seed(0)
    model = Uno3D_T20(6, 4, pad=3).to(dev())
    opt = ComplexAdam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    x, y = torch.randn(2, 32, 32, 10, 1, device=dev()), torch.randn(2, 32, 32, 20, device=dev())
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        ns3d_loss(model, x, y).backward()
        opt.step()
    assert redzone.check("NS-3D step") >= 50


@pytest.mark.parametrize("cfg", [(2, 3, 2, (16, 16, 10), (12, 12, 16), (4, 4, 3)), (1, 2, 2, (16, 16, 20), (8, 8, 20), (6, 6, 7)),
                                 (8, 8, 8, (32, 32, 13), (16, 16, 15), (6, 6, 5))])
def test_spectral_conv3d(redzone, cfg):
    from uno_amd.spectral3d import spectral_conv3d
    B, Ci, Co, din, dout, modes = cfg
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, Ci, *din, generator=g).to(dev()).requires_grad_(True)
    ws = [torch.randn(Ci, Co, *modes, dtype=torch.cfloat, generator=g).to(dev()).requires_grad_(True) for _ in range(4)]
    y = spectral_conv3d(x, ws, *dout)
    y.backward(torch.randn(B, Co, *dout, generator=g).to(dev()))
    assert redzone.check(f"spectral_conv3d {cfg}") >= 4


# ---------------------------------------------------------------------------------------------------------------- read side
# The guards above see stores.  Out-of-bounds READS (clamped raw-buffer loads, rows past a tile multiplied by zero weights) are checked
# from the other side: every input - activations, weights, gradients - lives between two 64 KiB runs of NaN.  A kernel that reads a
# neighbouring value AND lets it reach a result (0 x NaN = NaN) produces a non-finite or different output; results must equal the
# run on ordinary tensors bit for bit (all kernels are deterministic).
def nan_wrapped(t):
    """a copy of device tensor t whose storage is preceded and followed by 64 KiB of NaN"""
    flat = (torch.view_as_real(t) if t.is_complex() else t).detach().contiguous().reshape(-1)
    pad = GUARD // flat.element_size()
    raw = torch.full((pad + flat.numel() + pad,), float("nan"), dtype=flat.dtype, device=t.device)
    raw[pad:pad + flat.numel()] = flat
    v = raw[pad:pad + flat.numel()]
    v = torch.view_as_complex(v.view(*t.shape, 2)) if t.is_complex() else v.view(t.shape)
    return v.requires_grad_(t.requires_grad)


def _same(a, b, what):
    for x, y in zip(a, b):
        assert bool(torch.isfinite(torch.view_as_real(x) if x.is_complex() else x.float()).all()), f"{what}: non-finite result"
        assert torch.equal(x, y), f"{what}: a value outside an input reached the result"


@pytest.mark.parametrize("cfg", SPECTRAL_2D)
def test_reads_stay_inside_inputs_spectral_conv2d(cfg):
    from uno_amd.integral_operators import spectral_conv2d
    B, Ci, Co, H, W, Ho, Wo, m1, m2 = cfg
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(B, Ci, H, W, generator=g).to(dev()), torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g).to(dev()),
            torch.randn(Ci, Co, m1, m2, dtype=torch.cfloat, generator=g).to(dev())]
    gy = torch.randn(B, Co, Ho, Wo, generator=g).to(dev())
    res = []
    for wrap in (False, True):
        x, w1, w2 = ((nan_wrapped(t.requires_grad_(True)) if wrap else t.clone().requires_grad_(True)) for t in base)
        y = spectral_conv2d(x, w1, w2, Ho, Wo)
        y.backward(nan_wrapped(gy) if wrap else gy)
        res.append([y.detach(), x.grad, w1.grad, w2.grad])
    _same(res[1], res[0], f"spectral_conv2d {cfg}")


@pytest.mark.parametrize("cfg", [(2, 3, 2, (16, 16, 10), (12, 12, 16), (4, 4, 3)), (8, 8, 8, (32, 32, 13), (16, 16, 15), (6, 6, 5)),
                                 (7, 7, 8, (64, 64, 20), (48, 48, 20), (16, 16, 8))])
def test_reads_stay_inside_inputs_spectral_conv3d(cfg):
    from uno_amd.spectral3d import spectral_conv3d
    B, Ci, Co, din, dout, modes = cfg
    g = torch.Generator().manual_seed(4)
    base = [torch.randn(B, Ci, *din, generator=g).to(dev())] + [torch.randn(Ci, Co, *modes, dtype=torch.cfloat, generator=g).to(dev()) for _ in range(4)]
    gy = torch.randn(B, Co, *dout, generator=g).to(dev())
    res = []
    for wrap in (False, True):
        x, *ws = ((nan_wrapped(t.requires_grad_(True)) if wrap else t.clone().requires_grad_(True)) for t in base)
        y = spectral_conv3d(x, ws, *dout)
        y.backward(nan_wrapped(gy) if wrap else gy)
        res.append([y.detach(), x.grad] + [w.grad for w in ws])
    _same(res[1], res[0], f"spectral_conv3d {cfg}")


def _wrap_module(m):
    for p in m.parameters():
        p.data = nan_wrapped(p.data)
    return m


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("geom", [(37, 37, 55, 55), (55, 55, 37, 37), (70, 66, 35, 33), (223, 223, 111, 111)])
def test_reads_stay_inside_inputs_operator_block_2d(geom, normalize):
    """spectral branch + banded resampling (K7: rows past a tile meet zero weights) + 1x1 layer and its weight gradient + norm"""
    from uno_amd.integral_operators import OperatorBlock_2D
    H, W, Ho, Wo = geom
    res = []
    for wrap in (False, True):
        torch.manual_seed(3)
        blk = OperatorBlock_2D(6, 5, Ho, Wo, 5, 6, Normalize=normalize).to(dev())
        x = torch.randn(2, 6, H, W, device=dev())
        if wrap:
            blk, x = _wrap_module(blk), nan_wrapped(x)
        x.requires_grad_(True)
        y = blk(x)
        y.square().sum().backward()
        res.append([y.detach(), x.grad] + [p.grad for p in blk.parameters()])
    _same(res[1], res[0], f"OperatorBlock_2D {geom} normalize={normalize}")


def test_reads_stay_inside_inputs_training_steps():
    """two Darcy steps (S = 75, ragged levels) and two NS-3D steps with every parameter, input and target between NaN runs"""
    from uno_amd.harness import ComplexAdam, DarcyTrainer, UNO_9, Uno3D_T20, ns3d_loss, synthetic_darcy_batch
    res = []
    for wrap in (False, True):
        torch.manual_seed(0)
        model = UNO_9(3, 16, pad=5).to(dev())
        a, u = synthetic_darcy_batch(2, 75, 5, dev())
        if wrap:
            model, a, u = _wrap_module(model), nan_wrapped(a), nan_wrapped(u)
        tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
        losses = [tr.step(a, u) for _ in range(2)]
        res.append(losses + [p.detach().clone() for p in model.parameters()])
    _same(res[1], res[0], "Darcy steps")
    res = []
    for wrap in (False, True):
        torch.manual_seed(0)
        model = Uno3D_T20(6, 4, pad=3).to(dev())
        x, y = torch.randn(2, 32, 32, 10, 1, device=dev()), torch.randn(2, 32, 32, 20, device=dev())
        if wrap:
            model, x, y = _wrap_module(model), nan_wrapped(x), nan_wrapped(y)
        opt = ComplexAdam(model.parameters(), lr=1e-3, weight_decay=1e-3)
        out = []
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss = ns3d_loss(model, x, y)
            loss.backward()
            opt.step()
            out.append(loss.detach())
        res.append(out + [p.detach().clone() for p in model.parameters()])
    _same(res[1], res[0], "NS-3D steps")


# ---- round 5: the lift kernels (virtual first layer, K15) and the windowed fc1 - GELU - fc2 path need widths >= 260, which the ragged
# S = 75 steps above never reach
def _lift_and_window_pass(wrap):
    from uno_amd.integral_operators import channel_mix_cat_project, lift_gelu_pad
    torch.manual_seed(21)
    H, W, pad = 9, 261, 3
    f1, f0 = torch.nn.Linear(3, 32).to(dev()), torch.nn.Linear(32, 64).to(dev())
    fc1, fc2 = torch.nn.Linear(128, 64).to(dev()), torch.nn.Linear(64, 1).to(dev())
    x = torch.randn(2, 3, H, W, device=dev())
    c5 = torch.randn(2, 64, H + pad, W + pad, device=dev(), requires_grad=True)
    gout = torch.randn(2, 1, H, W, device=dev())
    if wrap:
        f1, f0, fc1, fc2 = (_wrap_module(m) for m in (f1, f0, fc1, fc2))
        x, c5, gout = nan_wrapped(x), nan_wrapped(c5), nan_wrapped(gout)
    lifted = lift_gelu_pad(x, f1, f0, pad, pad)
    out = channel_mix_cat_project([c5, lifted], fc1.weight, fc1.bias, fc2.weight, fc2.bias, gelu_first=True, crop=(H, W))[:, :, :H, :W]
    out.backward(gout)
    return [out.detach(), c5.grad] + [p.grad for m in (f1, f0, fc1, fc2) for p in m.parameters()]


def test_lift_and_window_kernels(redzone):
    res = _lift_and_window_pass(False)
    assert all(bool(torch.isfinite(t).all()) for t in res)
    assert redzone.check("lift + windowed projection") >= 8


def test_reads_stay_inside_inputs_lift_and_window():
    _same(_lift_and_window_pass(True), _lift_and_window_pass(False), "lift + windowed projection")
