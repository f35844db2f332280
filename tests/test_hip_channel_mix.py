"""K8 / K9 (csrc/channel_mix.hip) against torch in float64: the 1x1 convolution of pointwise_op_2D/3D (reference
integral_operators.py:219, 439) and the channels-first lift / projection layers - forward, input gradient,
weight and bias gradient.  f32 accumulation: tolerance 2e-6 (l2-relative) for the channel sums, 2e-5 for the
pixel-long reductions of the weight / bias gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # B, Ci, Co, P
    (1, 1, 1, 1),
    (2, 3, 5, 17),
    (2, 16, 16, 128),
    (3, 64, 64, 1000),
    (2, 3, 32, 431 * 431),        # lift of the Darcy model (fc0)
    (2, 64, 128, 215 * 215),
    (2, 256, 256, 2500),
    (4, 100, 36, 4097),
    (1, 128, 1, 777),             # final projection
    (5, 20, 70, 33),
    (3, 1, 5, 1030), (2, 2, 16, 2048), (2, 4, 33, 4099), (2, 3, 17, 1025),     # few input channels: the streaming forms of K8 / K9
]


def rel(a, b):
    d = (a.double() - b).norm().item()
    n = b.norm().item()
    return d / n if n > 0 else d


def _ref(x, w, b):
    y = torch.matmul(w.double(), x.double())
    if b is not None:
        y = y + b.double().view(1, -1, 1)
    return y


@pytest.mark.parametrize("B,Ci,Co,P", SHAPES)
@pytest.mark.parametrize("with_bias", [True, False])
def test_forward(B, Ci, Co, P, with_bias):
    from uno_amd import _native
    g = torch.Generator().manual_seed(B * 1000 + Ci * 10 + Co)
    x = torch.randn(B, Ci, P, generator=g).cuda()
    w = torch.randn(Co, Ci, generator=g).cuda()
    b = torch.randn(Co, generator=g).cuda() if with_bias else None
    y = _native.channel_mix(x, w, b)
    assert rel(y, _ref(x, w, b)) < 2e-6


@pytest.mark.parametrize("B,Ci,Co,P", SHAPES)
def test_transposed(B, Ci, Co, P):
    from uno_amd import _native
    g = torch.Generator().manual_seed(7 + Ci)
    gy = torch.randn(B, Co, P, generator=g).cuda()
    w = torch.randn(Co, Ci, generator=g).cuda()
    gx = _native.channel_mix(gy, w, None, transpose_w=True)
    assert gx.shape == (B, Ci, P)
    assert rel(gx, torch.matmul(w.double().t(), gy.double())) < 2e-6


@pytest.mark.parametrize("B,Ci,Co,P", SHAPES)
def test_wgrad(B, Ci, Co, P):
    from uno_amd import _native
    g = torch.Generator().manual_seed(11 + Co)
    gy = torch.randn(B, Co, P, generator=g).cuda()
    x = torch.randn(B, Ci, P, generator=g).cuda()
    gw, gb = _native.channel_wgrad(gy, x)
    ref_w = torch.einsum("bop,bip->oi", gy.double(), x.double())
    ref_b = gy.double().sum(dim=(0, 2))
    assert rel(gw, ref_w) < 2e-5
    assert rel(gb, ref_b) < 2e-5
    gw2, none = _native.channel_wgrad(gy, x, need_bias=False)
    assert none is None and torch.equal(gw, gw2)          # fixed-order reduction: bit-reproducible


def test_autograd_matches_conv2d():
    from uno_amd.integral_operators import channel_mix
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(24, 40, 1).cuda()
    x = torch.randn(3, 24, 37, 41, device="cuda", requires_grad=True)
    y = channel_mix(x, conv.weight, conv.bias)
    gy = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (x, conv.weight, conv.bias), gy)
    x2 = x.detach().double().requires_grad_(True)
    conv2 = torch.nn.Conv2d(24, 40, 1).cuda().double()
    conv2.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    y2 = conv2(x2)
    gx2, gw2, gb2 = torch.autograd.grad(y2, (x2, conv2.weight, conv2.bias), gy.double())
    assert gw.shape == conv.weight.shape
    for a, b in ((y, y2), (gx, gx2), (gw, gw2), (gb, gb2)):
        assert rel(a, b) < 1e-5


def test_empty_batch_and_errors():
    from uno_amd import _native
    y = _native.channel_mix(torch.zeros(0, 4, 9, device="cuda"), torch.zeros(6, 4, device="cuda"))
    assert y.shape == (0, 6, 9)
    gw, gb = _native.channel_wgrad(torch.zeros(0, 6, 9, device="cuda"), torch.zeros(0, 4, 9, device="cuda"))
    assert gw.abs().sum() == 0 and gb.abs().sum() == 0
    with pytest.raises(RuntimeError):
        _native.channel_mix(torch.zeros(1, 4, 9), torch.zeros(6, 4))               # CPU tensors
    with pytest.raises(RuntimeError):
        _native.channel_mix(torch.zeros(1, 5, 9, device="cuda"), torch.zeros(6, 4, device="cuda"))


def test_accumulate():
    from uno_amd import _native
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 20, 1000, generator=g).cuda()
    w = torch.randn(12, 20, generator=g).cuda()
    b = torch.randn(12, generator=g).cuda()
    base = torch.randn(3, 12, 1000, generator=g).cuda()
    out = base.clone()
    ret = _native.channel_mix(x, w, b, out=out)
    assert ret.data_ptr() == out.data_ptr()
    assert rel(out, base.double() + _ref(x, w, b)) < 2e-6
    gbase = torch.randn(3, 20, 1000, generator=g).cuda()
    gout = gbase.clone()
    gy = torch.randn(3, 12, 1000, generator=g).cuda()
    _native.channel_mix(gy, w, None, transpose_w=True, out=gout)
    assert rel(gout, gbase.double() + torch.matmul(w.double().t(), gy.double())) < 2e-6
    with pytest.raises(RuntimeError):
        _native.channel_mix(x, w, b, out=torch.zeros(3, 11, 1000, device="cuda"))


def test_channel_mix_cat_equals_mix_of_cat():
    from uno_amd.integral_operators import channel_mix, channel_mix_cat
    torch.manual_seed(2)
    lin = torch.nn.Linear(24 + 40, 36).cuda()
    a = torch.randn(3, 24, 19, 23, device="cuda", requires_grad=True)
    b = torch.randn(3, 40, 19, 23, device="cuda", requires_grad=True)
    y = channel_mix_cat([a, b], lin.weight, lin.bias)
    gy = torch.randn_like(y)
    got = torch.autograd.grad(y, (a, b, lin.weight, lin.bias), gy)
    y2 = channel_mix(torch.cat([a, b], dim=1), lin.weight, lin.bias)
    ref = torch.autograd.grad(y2, (a, b, lin.weight, lin.bias), gy)
    assert rel(y, y2.double()) < 2e-6
    for g, r in zip(got, ref):
        assert g.shape == r.shape and rel(g, r.double()) < 2e-5


@pytest.mark.parametrize("B,Ci,Co,shape", [(2, 32, 64, (45, 41)), (3, 7, 5, (19,)), (1, 16, 130, (300,)), (2, 64, 64, (3,))])
def test_gelu_channel_mix_matches_stock_sequence(B, Ci, Co, shape):
    """y = W gelu(pre) + b with the GELU applied on read (K8 act_in, K9 act_x) and gelu'(pre) in the input-gradient epilogue,
    against F.gelu -> channel mix in float64."""
    from uno_amd.integral_operators import gelu_channel_mix
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(Ci + Co)
    pre = (1.5 * torch.randn(B, Ci, *shape, generator=g)).cuda().requires_grad_(True)
    w = torch.randn(Co, Ci, generator=g).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda().requires_grad_(True)
    y = gelu_channel_mix(pre, w, b)
    gy = torch.randn_like(y)
    got = torch.autograd.grad(y, (pre, w, b), gy)
    pre2, w2, b2 = (t.detach().double().requires_grad_(True) for t in (pre, w, b))
    y2 = torch.einsum("oc,bc...->bo...", w2, F.gelu(pre2)) + b2.view(1, -1, *([1] * len(shape)))
    ref = torch.autograd.grad(y2, (pre2, w2, b2), gy.double())
    assert rel(y, y2.detach()) < 2e-6
    assert rel(got[0], ref[0]) < 2e-6
    assert rel(got[1], ref[1]) < 2e-5 and rel(got[2], ref[2]) < 2e-5


def test_channel_mix_cat_with_deferred_gelu():
    from uno_amd.integral_operators import channel_mix_cat
    import torch.nn.functional as F
    torch.manual_seed(4)
    lin = torch.nn.Linear(24 + 40, 36).cuda()
    a = (1.5 * torch.randn(3, 24, 19, 23, device="cuda")).requires_grad_(True)
    b = torch.randn(3, 40, 19, 23, device="cuda", requires_grad=True)
    y = channel_mix_cat([a, b], lin.weight, lin.bias, gelu_first=True)
    gy = torch.randn_like(y)
    got = torch.autograd.grad(y, (a, b, lin.weight, lin.bias), gy)
    a2, b2 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    w2, c2 = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    y2 = torch.einsum("oc,bchw->bohw", w2, torch.cat([F.gelu(a2), b2], dim=1)) + c2.view(1, -1, 1, 1)
    ref = torch.autograd.grad(y2, (a2, b2, w2, c2), gy.double())
    assert rel(y, y2.detach()) < 2e-6
    for g_, r_ in zip(got, ref):
        assert g_.shape == r_.shape and rel(g_, r_) < 2e-5


# ------------------------------------------------------------------ two sources / two destinations / activated second output
TWO = [
    # B, C1, C2, Co, P
    (2, 64, 64, 64, 446 * 9 + 3),      # fc1 of the Darcy model (cat([conv5 out, lifted]) -> 64), ragged last pixel tile
    (2, 128, 128, 64, 223 * 7),        # conv5's 1x1 convolution on cat([conv4 out, c0])
    (3, 16, 48, 40, 515),              # split at one 16-channel chunk, Co not a tile multiple
    (2, 32, 16, 128, 700),             # wide (128-channel) kernel with two sources
    (1, 64, 192, 256, 130),
    (2, 48, 20, 33, 77),               # second source with a ragged chunk (guarded path)
    (2, 64, 64, 64, 3),                # rows shorter than 4 pixels: scalar path
]


def _gelu64(t):
    return torch.nn.functional.gelu(t.double())


@pytest.mark.parametrize("B,C1,C2,Co,P", TWO)
@pytest.mark.parametrize("act_in", [False, True])
def test_two_source_forward(B, C1, C2, Co, P, act_in):
    from uno_amd import _native
    g = torch.Generator().manual_seed(C1 + 3 * C2 + Co + P)
    x1, x2 = torch.randn(B, C1, P, generator=g).cuda(), torch.randn(B, C2, P, generator=g).cuda()
    w, b = torch.randn(Co, C1 + C2, generator=g).cuda(), torch.randn(Co, generator=g).cuda()
    y = _native.channel_mix2(x1, x2, w, b, act_in=act_in)
    ref = _ref(torch.cat([_gelu64(x1) if act_in else x1.double(), x2.double()], 1), w, b)
    assert rel(y, ref) < 2e-6
    base = torch.randn(B, Co, P, generator=g).cuda()
    out = base.clone()
    _native.channel_mix2(x1, x2, w, b, act_in=act_in, out=out, accumulate=True)
    assert rel(out, ref + base.double()) < 2e-6
    if not act_in:
        y2, act = _native.channel_mix2(x1, x2, w, b, y_act=True)
        assert torch.equal(y2, y) and rel(act, _gelu64(y)) < 2e-6
        y1, a1 = _native.channel_mix2(x1, None, w[:, :C1].contiguous(), b, y_act=True)       # single source + activated copy
        assert rel(y1, _ref(x1, w[:, :C1], b)) < 2e-6 and rel(a1, _gelu64(y1)) < 2e-6


@pytest.mark.parametrize("B,C1,C2,Co,P", [c for c in TWO if c[1] % 64 == 0])
@pytest.mark.parametrize("dgelu", [False, True])
def test_two_destination_input_gradients(B, C1, C2, Co, P, dgelu):
    """the transposed call on grad_y: channels [0, C1) of W^T gy (optionally * gelu'(pre)) and channels [C1, C1 + C2) land in two
    tensors from one pass; the accumulating form adds into both"""
    from uno_amd import _native
    g = torch.Generator().manual_seed(C1 + C2 + Co)
    gy = torch.randn(B, Co, P, generator=g).cuda()
    w = torch.randn(Co, C1 + C2, generator=g).cuda()
    pre = torch.randn(B, C1, P, generator=g).cuda() if dgelu else None
    g1, g2 = _native.channel_mix2(gy, None, w, None, transpose_w=True, split_out=C1, dgelu_of=pre)
    ref = torch.matmul(w.double().t(), gy.double())
    r1, r2 = ref[:, :C1], ref[:, C1:]
    if dgelu:
        pd = pre.double().requires_grad_(True)
        torch.nn.functional.gelu(pd).sum().backward()
        r1 = r1 * pd.grad
    assert g1.shape == (B, C1, P) and g2.shape == (B, C2, P)
    assert rel(g1, r1) < 2e-6 and rel(g2, r2) < 2e-6
    if not dgelu:
        b1, b2 = torch.randn(B, C1, P, generator=g).cuda(), torch.randn(B, C2, P, generator=g).cuda()
        o1, o2 = b1.clone(), b2.clone()
        _native.channel_mix2(gy, None, w, None, transpose_w=True, out=o1, out2=o2, split_out=C1, accumulate=True)
        assert rel(o1, r1 + b1.double()) < 2e-6 and rel(o2, r2 + b2.double()) < 2e-6


@pytest.mark.parametrize("B,C1,C2,Co,P", [c for c in TWO if c[1] % 64 == 0 and c[4] >= 64])
@pytest.mark.parametrize("act_x", [False, True])
def test_two_source_wgrad(B, C1, C2, Co, P, act_x):
    from uno_amd import _native
    g = torch.Generator().manual_seed(C1 + C2 + Co + 1)
    gy = torch.randn(B, Co, P, generator=g).cuda()
    x1, x2 = torch.randn(B, C1, P, generator=g).cuda(), torch.randn(B, C2, P, generator=g).cuda()
    gw, gb = _native.channel_wgrad2(gy, x1, x2, act_x=act_x)
    xc = torch.cat([_gelu64(x1) if act_x else x1.double(), x2.double()], 1)
    assert gw.shape == (Co, C1 + C2)
    assert rel(gw, torch.einsum("bop,bip->oi", gy.double(), xc)) < 2e-5
    assert rel(gb, gy.double().sum(dim=(0, 2))) < 2e-5


@pytest.mark.parametrize("B,Ci,C2,Co,P,act_x", [(2, 8, 0, 8, 576, False), (3, 64, 64, 64, 300, True), (2, 3, 0, 32, 2048, False), (1, 192, 0, 96, 77, False)])
def test_wgrad_first_stage_then_one_second_stage(B, Ci, C2, Co, P, act_x):
    """uno_channel_wgrad2 with accumulate = 3 leaves its split-K partial sums; uno_channel_wgrad_finish sums the partial sums of
    SEVERAL calls (the uses of a layer in a roll-out, reference ns_train_2d.py:46-68) in one second stage: same result as the sum of
    the separate weight gradients, written or accumulated."""
    from uno_amd import _native
    g = torch.Generator().manual_seed(B + Ci + Co + P)
    T = 3
    gys = [torch.randn(B, Co, P, generator=g).cuda() for _ in range(T)]
    x1s = [torch.randn(B, Ci, P, generator=g).cuda() for _ in range(T)]
    x2s = [torch.randn(B, C2, P, generator=g).cuda() if C2 else None for _ in range(T)]
    nf = _native.channel_wgrad_partial_floats(B, Ci + C2, Co, P)
    assert nf > 0 and nf % (Co * (Ci + C2 + 1)) == 0
    parts = torch.full((T, nf), float("nan"), device="cuda")            # every float of a row must be written by its call
    refw = torch.zeros(Co, Ci + C2, dtype=torch.float64, device="cuda")
    refb = torch.zeros(Co, dtype=torch.float64, device="cuda")
    for t in range(T):
        assert _native.channel_wgrad2(gys[t], x1s[t], x2s[t], act_x=act_x, partials_out=parts[t]) == (None, None)
        gw, gb = _native.channel_wgrad2(gys[t], x1s[t], x2s[t], act_x=act_x)
        refw += gw.double(); refb += gb.double()
    gw, gb = _native.channel_wgrad_finish(parts, Ci + C2, Co, True)
    assert rel(gw, refw) < 2e-6 and rel(gb, refb) < 2e-6
    bw, bb = torch.randn(Co, Ci + C2, generator=g).cuda(), torch.randn(Co, generator=g).cuda()
    ow, ob = bw.clone(), bb.clone()
    _native.channel_wgrad_finish(parts[1:], Ci + C2, Co, True, out_w=ow, out_b=ob, accumulate=True)
    gw1, _ = _native.channel_wgrad_finish(parts[:1], Ci + C2, Co, False)
    assert rel(ow + gw1, refw + bw.double()) < 2e-6
    with pytest.raises(RuntimeError):
        _native.channel_wgrad_finish(parts.flatten()[:-1], Ci + C2, Co, True)          # not a whole number of blocks
    with pytest.raises(RuntimeError):
        _native.channel_wgrad2(gys[0], x1s[0], x2s[0], partials_out=parts[0][:-4].contiguous())     # wrong size


def test_two_source_argument_errors():
    from uno_amd import _native
    x1, x2 = torch.randn(1, 24, 200).cuda(), torch.randn(1, 8, 200).cuda()
    w = torch.randn(64, 32).cuda()
    with pytest.raises(RuntimeError):            # sources split inside a 16-channel chunk
        _native.channel_mix2(x1, x2, w, None)
    gy = torch.randn(1, 64, 200).cuda()
    with pytest.raises(RuntimeError):            # destinations split inside a 64-channel tile
        _native.channel_mix2(gy, None, torch.randn(64, 100).cuda(), None, transpose_w=True, split_out=40)


@pytest.mark.parametrize("C1,C2", [(64, 64), (24, 8), (128, 64)])
@pytest.mark.parametrize("gelu_first", [False, True])
def test_channel_mix_cat_autograd_vs_torch(C1, C2, gelu_first):
    """channel_mix_cat (fused two-source kernels where the split rules allow, two accumulating calls otherwise) against
    F.conv1d on the concatenation in float64: output, both input gradients, weight and bias gradients."""
    from uno_amd.integral_operators import channel_mix_cat
    torch.manual_seed(C1 + C2)
    B, Co, P = 2, 64, 1000
    x1 = torch.randn(B, C1, P).cuda().requires_grad_(True)
    x2 = torch.randn(B, C2, P).cuda().requires_grad_(True)
    w = torch.randn(Co, C1 + C2).cuda().requires_grad_(True)
    b = torch.randn(Co).cuda().requires_grad_(True)
    gy = torch.randn(B, Co, P).cuda()
    y = channel_mix_cat([x1, x2], w, b, gelu_first=gelu_first)
    y.backward(gy)
    xd1, xd2, wd, bd = (t.detach().double().requires_grad_(True) for t in (x1, x2, w, b))
    a1 = torch.nn.functional.gelu(xd1) if gelu_first else xd1
    yr = torch.matmul(wd, torch.cat([a1, xd2], 1)) + bd.view(1, -1, 1)
    yr.backward(gy.double())
    assert rel(y.detach(), yr.detach()) < 2e-6
    assert rel(x1.grad, xd1.grad) < 2e-6 and rel(x2.grad, xd2.grad) < 2e-6
    assert rel(w.grad, wd.grad) < 2e-5 and rel(b.grad, bd.grad) < 2e-5


@pytest.mark.parametrize("B,C1,C2,Co,P", [(2, 64, 64, 64, 446 * 9 + 3), (2, 16, 48, 40, 515), (1, 32, 0, 64, 128 * 5), (2, 64, 64, 33, 77), (2, 16, 16, 16, 3)])
@pytest.mark.parametrize("act_in", [False, True])
def test_fused_projection(B, C1, C2, Co, P, act_in):
    """uno_channel_mix2 with proj_w: proj[b, p] = b2 + sum_o w2[o] gelu(y[b, o, p]) from the pass that writes y (interior tiles,
    ragged pixel tiles, partial channel tiles, the scalar path)."""
    from uno_amd import _native
    g = torch.Generator().manual_seed(C1 + C2 + Co + P)
    x1 = torch.randn(B, C1, P, generator=g).cuda()
    x2 = torch.randn(B, C2, P, generator=g).cuda() if C2 else None
    w, b = (torch.randn(Co, C1 + C2, generator=g) / (C1 + C2) ** 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    w2, b2 = torch.randn(Co, generator=g).cuda(), torch.randn(1, generator=g).cuda()
    y, proj = _native.channel_mix2(x1, x2, w, b, act_in=act_in, project=(w2, b2))
    xs = [_gelu64(x1) if act_in else x1.double()] + ([x2.double()] if C2 else [])
    yr = _ref(torch.cat(xs, 1), w, b)
    pr = (w2.double().view(1, -1, 1) * torch.nn.functional.gelu(yr)).sum(1) + b2.double()
    assert rel(y, yr) < 2e-6 and proj.shape == (B, P) and rel(proj, pr) < 3e-6


def test_channel_mix_cat_project_autograd_vs_torch():
    from uno_amd.integral_operators import channel_mix_cat_project
    torch.manual_seed(4)
    B, C1, C2, Co, P = 2, 64, 64, 64, 1000
    x1 = torch.randn(B, C1, P).cuda().requires_grad_(True)
    x2 = torch.randn(B, C2, P).cuda().requires_grad_(True)
    w = (torch.randn(Co, C1 + C2) / 11).cuda().requires_grad_(True)
    b = torch.randn(Co).cuda().requires_grad_(True)
    w2 = torch.randn(1, Co).cuda().requires_grad_(True)
    b2 = torch.randn(1).cuda().requires_grad_(True)
    gout = torch.randn(B, 1, P).cuda()
    out = channel_mix_cat_project([x1, x2], w, b, w2, b2, gelu_first=True)
    out.backward(gout)
    d = [t.detach().double().requires_grad_(True) for t in (x1, x2, w, b, w2, b2)]
    yr = torch.matmul(d[2], torch.cat([torch.nn.functional.gelu(d[0]), d[1]], 1)) + d[3].view(1, -1, 1)
    outr = torch.matmul(d[4], torch.nn.functional.gelu(yr)) + d[5].view(1, 1, 1)
    outr.backward(gout.double())
    assert rel(out.detach(), outr.detach()) < 3e-6
    for got, ref, tol in zip((x1, x2, w, b, w2, b2), d, (3e-6, 3e-6, 2e-5, 2e-5, 2e-5, 2e-5)):
        assert rel(got.grad, ref.grad) < tol


# ---- K8-S: the wide layers (Ci >= 128, Co % 128 == 0) on the bf16 matrix pipe with three-piece operands (csrc/channel_mix.hip).
# The kernel name is checked through the library's own launch log so that a dispatch change cannot leave these shapes on the f32 form.
SPLIT = [
    # B, C1, C2, Co, P
    (2, 128, 0, 256, 1111),         # 8 full pixel tiles + a partial one (guarded fallback inside the kernel)
    (1, 256, 0, 128, 640),
    (2, 160, 0, 128, 300),          # five chunks of 32
    (2, 96, 64, 128, 515),          # two sources, split on a chunk boundary
    (2, 128, 128, 256, 111 * 111),  # the Darcy model's 111^2 level (conv2: 256 -> 256)
]


def _launched(fn):
    from uno_amd import _native
    _native.profile_begin(64)
    out = fn()
    torch.cuda.synchronize()
    return out, [r[0] for r in _native.profile_end()]


@pytest.mark.parametrize("B,C1,C2,Co,P", SPLIT)
def test_split_bf16_wide_layers(B, C1, C2, Co, P):
    from uno_amd import _native
    g = torch.Generator().manual_seed(C1 + 5 * C2 + Co + P)
    # values spread over several binades and signs: the three-piece split must carry the low bits of every element
    x = (torch.randn(B, C1 + C2, P, generator=g) * torch.exp2(torch.randint(-6, 7, (B, C1 + C2, 1), generator=g).float())).cuda()
    w = (torch.randn(Co, C1 + C2, generator=g) * torch.exp2(torch.randint(-4, 5, (Co, 1), generator=g).float())).cuda()
    b = torch.randn(Co, generator=g).cuda()
    x1, x2 = (x[:, :C1].contiguous(), x[:, C1:].contiguous()) if C2 else (x, None)
    ref = _ref(x, w, b)
    y, names = _launched(lambda: _native.channel_mix2(x1, x2, w, b))
    assert names == ["uno::channel_mix_split_kernel"], names
    assert rel(y, ref) < 1e-6          # (measured ~1e-7: tighter than the f32 MFMA form's bound on purpose)
    assert (y.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    # accumulate + activated copy
    base = torch.randn(B, Co, P, generator=g).cuda()
    out = base.clone()
    _native.channel_mix2(x1, x2, w, b, out=out, accumulate=True)
    assert rel(out, ref + base.double()) < 1e-6
    y2, act = _native.channel_mix2(x1, x2, w, b, y_act=True)
    assert torch.equal(y2, y) and rel(act, _gelu64(y)) < 2e-6
    # transposed weights (the input-gradient call): gx = W^T gy, both destinations of a two-source layer when the split is 128-aligned
    gy = torch.randn(B, Co, P, generator=g).cuda()
    if (C1 + C2) % 128 == 0 and Co >= 128:
        gx, names = _launched(lambda: _native.channel_mix(gy, w, None, transpose_w=True))
        assert names == ["uno::channel_mix_split_kernel"], names
        assert rel(gx, torch.matmul(w.double().t(), gy.double())) < 1e-6
        if C2 and C1 % 128 == 0:
            g1, g2 = _native.channel_mix2(gy, None, w, None, transpose_w=True, split_out=C1)
            assert rel(torch.cat([g1, g2], 1), torch.matmul(w.double().t(), gy.double())) < 1e-6


SPLIT64 = [  # B, C1, C2, Co, P: 64-channel tiles of K8-S (Co % 64 == 0, not a multiple of 128)
    (2, 256, 0, 64, 1111), (1, 128, 0, 192, 640), (2, 160, 0, 64, 300), (2, 96, 64, 64, 515), (2, 128, 128, 64, 223 * 223 // 7),
]


@pytest.mark.parametrize("B,C1,C2,Co,P", SPLIT64)
def test_split_bf16_64_channel_tiles(B, C1, C2, Co, P):
    """conv5's 256 -> 64, the input gradients of the 64-channel levels: four waves = four pixel quarters x 64 channels"""
    from uno_amd import _native
    g = torch.Generator().manual_seed(C1 + 5 * C2 + Co + P + 1)
    x = (torch.randn(B, C1 + C2, P, generator=g) * torch.exp2(torch.randint(-6, 7, (B, C1 + C2, 1), generator=g).float())).cuda()
    w = (torch.randn(Co, C1 + C2, generator=g) * torch.exp2(torch.randint(-4, 5, (Co, 1), generator=g).float())).cuda()
    b = torch.randn(Co, generator=g).cuda()
    x1, x2 = (x[:, :C1].contiguous(), x[:, C1:].contiguous()) if C2 else (x, None)
    ref = _ref(x, w, b)
    y, names = _launched(lambda: _native.channel_mix2(x1, x2, w, b))
    assert names == ["uno::channel_mix_split_kernel"], names
    assert rel(y, ref) < 1e-6
    assert (y.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    base = torch.randn(B, Co, P, generator=g).cuda()
    out = base.clone()
    _native.channel_mix2(x1, x2, w, b, out=out, accumulate=True)
    assert rel(out, ref + base.double()) < 1e-6
    y2, act = _native.channel_mix2(x1, x2, w, b, y_act=True)
    assert torch.equal(y2, y) and rel(act, _gelu64(y)) < 2e-6
    # transposed weights with two 64-channel destinations (fc1's input gradients: 64 -> 64 + 64 needs Ci >= 128, so a 128-channel gy)
    gy = torch.randn(B, 128, P, generator=g).cuda()
    wt = torch.randn(128, 128, generator=g).cuda()
    (g1, g2), names = _launched(lambda: _native.channel_mix2(gy, None, wt, None, transpose_w=True, split_out=64))
    assert names == ["uno::channel_mix_split_kernel"], names
    assert rel(torch.cat([g1, g2], 1), torch.matmul(wt.double().t(), gy.double())) < 1e-6
    # bf16 activations
    xb = x.to(torch.bfloat16)
    yb, names = _launched(lambda: _native.channel_mix(xb, w, b))
    assert names == ["uno::channel_mix_split_kernel"], names
    assert rel(yb.float(), _ref(xb.float(), w, b)) < 3e-3


@pytest.mark.parametrize("bf", [False, True])
@pytest.mark.parametrize("Co", [64, 128, 192])
def test_split_bf16_gelu_on_read_and_projection(Co, bf):
    """K8-S with the GELU applied to the first source as it is staged (fc1 behind conv5: darcy_flow_uno2d.py:126-129) and, on a
    64-channel layer, the fused one-channel projection fc2(gelu(y))"""
    from uno_amd import _native
    B, C1, C2, P = 2, 128, 64, 128 * 5 + 37
    g = torch.Generator().manual_seed(Co + 7)
    x1, x2 = torch.randn(B, C1, P, generator=g).cuda(), torch.randn(B, C2, P, generator=g).cuda()
    w = (torch.randn(Co, C1 + C2, generator=g) / 14.0).cuda()
    b = torch.randn(Co, generator=g).cuda()
    if bf:
        x1, x2 = x1.to(torch.bfloat16), x2.to(torch.bfloat16)
    ref = _ref(torch.cat([_gelu64(x1.float()), x2.double()], 1), w, b)
    tol = 4e-3 if bf else 2e-6
    y, names = _launched(lambda: _native.channel_mix2(x1, x2, w, b, act_in=True))
    assert names == ["uno::channel_mix_split_kernel"], names
    assert rel(y.float(), ref) < tol
    if Co == 64:
        w2, b2 = torch.randn(Co, generator=g).cuda(), torch.randn(1, generator=g).cuda()
        (y2, proj), names = _launched(lambda: _native.channel_mix2(x1, x2, w, b, act_in=True, project=(w2, b2)))
        assert names == ["uno::channel_mix_split_kernel"], names
        assert torch.equal(y2, y)
        pref = b2.double() + torch.einsum("o,bop->bp", w2.double(), _gelu64(ref))
        assert rel(proj.float(), pref) < (1e-2 if bf else 5e-6)


@pytest.mark.parametrize("bf", [False, True])
@pytest.mark.parametrize("Co,Co1", [(192, 64), (256, 128), (128, 128), (64, 64)])
def test_split_bf16_gelu_derivative_epilogue(Co, Co1, bf):
    """the input-gradient call of a layer behind a fused-GELU block: gelu'(pre) on the first destination, written or accumulated, and in
    its 'completed sum' form (accumulate = 2: (old + product) * gelu')"""
    from uno_amd import _native
    B, Ci, P = 2, 128, 128 * 4 + 19
    g = torch.Generator().manual_seed(Co + Co1)
    gy = torch.randn(B, Ci, P, generator=g).cuda()
    w = (torch.randn(Ci, Co, generator=g) / 11.0).cuda()
    pre = torch.randn(B, Co1, P, generator=g).cuda()
    if bf:
        gy, pre = gy.to(torch.bfloat16), pre.to(torch.bfloat16)
    tol = 4e-3 if bf else 2e-6
    full = torch.matmul(w.double().t(), gy.double())
    pd = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(pd).sum().backward()
    d = pd.grad
    if Co1 < Co:
        (g1, g2), names = _launched(lambda: _native.channel_mix2(gy, None, w, None, transpose_w=True, split_out=Co1, dgelu_of=pre))
        assert names == ["uno::channel_mix_split_kernel"], names
        assert rel(g1.float(), full[:, :Co1] * d) < tol and rel(g2.float(), full[:, Co1:]) < tol
    else:
        g1, names = _launched(lambda: _native.channel_mix(gy, w, None, transpose_w=True, dgelu_of=pre))
        assert names == ["uno::channel_mix_split_kernel"], names
        assert rel(g1.float(), full * d) < tol
        base = torch.randn(B, Co, P, generator=g).cuda().to(gy.dtype)
        out = base.clone()
        _native.channel_mix(gy, w, None, transpose_w=True, out=out, dgelu_of=pre)
        assert rel(out.float(), base.double() + full * d) < tol
        out = base.clone()
        _, names = _launched(lambda: _native.channel_mix(gy, w, None, transpose_w=True, out=out, dgelu_of=pre, dgelu_total=True))
        assert names == ["uno::channel_mix_split_kernel"], names
        assert rel(out.float(), (base.double() + full) * d) < tol


def test_split_bf16_narrow_inputs_on_bf16_activations():
    """bf16 activations: the bf16-MFMA form from 32 input channels on (the lift 32 -> 64, the 64-channel levels' input gradients)"""
    from uno_amd import _native
    g = torch.Generator().manual_seed(3)
    for Ci, Co in ((32, 64), (64, 128), (64, 64)):
        x = torch.randn(2, Ci, 1000, generator=g).cuda().to(torch.bfloat16)
        w, b = (torch.randn(Co, Ci, generator=g) / 6.0).cuda(), torch.randn(Co, generator=g).cuda()
        y, names = _launched(lambda: _native.channel_mix(x, w, b))
        assert names == ["uno::channel_mix_split_kernel"], names
        assert rel(y.float(), _ref(x.float(), w, b)) < 4e-3
        yf, names = _launched(lambda: _native.channel_mix(x.float(), w, b))
        assert names != ["uno::channel_mix_split_kernel"], names              # f32 activations: from 128 input channels on


def test_split_bf16_wide_layers_bf16_activations():
    from uno_amd import _native
    g = torch.Generator().manual_seed(11)
    B, Ci, Co, P = 2, 256, 128, 1000
    x = torch.randn(B, Ci, P, generator=g).cuda().to(torch.bfloat16)
    w, b = torch.randn(Co, Ci, generator=g).cuda(), torch.randn(Co, generator=g).cuda()
    y, names = _launched(lambda: _native.channel_mix(x, w, b))
    assert names == ["uno::channel_mix_split_kernel"], names
    ref = _ref(x.float(), w, b)
    # the products are exact to ~1e-7; what remains is the bf16 rounding of the result
    assert rel(y.float(), ref) < 3e-3
    assert rel(y.float(), ref.float().to(torch.bfloat16).double()) < 1e-3


# K9-S (channel_wgrad_split_kernel): both channel counts >= 96 - three-piece bf16 operands on the bf16 MFMA, 128 x 128 weight tiles.
# Shapes: full tiles, partial tiles in both directions, rows ending inside a 32-pixel half chunk, inside a 4-pixel group, pixel counts
# that leave the second half of the last 64-pixel chunk empty, two sources, GELU-on-read, the 64-row tiles (Co < 96).
WIDE_WGRAD = [  # B, C1, C2, Co, P, act_x   (batch x pixels >= 100 000: below that the vector kernel keeps the layer)
    (2, 128, 0, 128, 51200, False), (2, 256, 0, 256, 50013, False), (2, 96, 0, 130, 50047, False), (1, 130, 0, 96, 100031, True),
    (2, 64, 32, 128, 50001, True), (4, 64, 32, 128, 25010, False), (1, 128, 128, 128, 100000 + 32, False), (2, 64, 64, 100, 50200, True),
    (1, 257, 0, 129, 100130, False), (2, 128, 0, 50, 50300, False), (1, 96, 0, 64, 100000 + 64 + 17, True), (3, 192, 0, 48, 33400 + 29, False),
]


@pytest.mark.parametrize("B,C1,C2,Co,P,act_x", WIDE_WGRAD)
def test_wide_wgrad_split_bf16(B, C1, C2, Co, P, act_x):
    from uno_amd import _native
    g = torch.Generator().manual_seed(C1 + C2 + Co + P)
    gy = torch.randn(B, Co, P, generator=g).cuda()
    x1 = torch.randn(B, C1, P, generator=g).cuda()
    x2 = torch.randn(B, C2, P, generator=g).cuda() if C2 else None
    _native.profile_begin(64)
    gw, gb = _native.channel_wgrad2(gy, x1, x2, act_x=act_x)
    torch.cuda.synchronize()
    names = {n for n, _, _ in _native.profile_end()}
    assert "uno::channel_wgrad_split_kernel" in names, names
    xs = [_gelu64(x1) if act_x else x1.double()] + ([x2.double()] if C2 else [])
    xc = torch.cat(xs, 1)
    assert gw.shape == (Co, C1 + C2)
    # the six-product split is MORE accurate than the f32 MFMA form (profiles/r04_split_bf16_error.txt): same bound as the narrow layers
    assert rel(gw, torch.einsum("bop,bip->oi", gy.double(), xc)) < 2e-6
    assert rel(gb, gy.double().sum(dim=(0, 2))) < 2e-6
    gw2, _ = _native.channel_wgrad2(gy, x1, x2, act_x=act_x)
    assert torch.equal(gw, gw2)                                     # fixed-order reduction: bit-reproducible


@pytest.mark.parametrize("B,C1,C2,Co,P,act_x", WIDE_WGRAD)
def test_wide_wgrad_split_bf16_activations(B, C1, C2, Co, P, act_x):
    """bf16 activations: both operands are exact in ONE bf16 piece (one product); with GELU-on-read, gelu(x) is an f32 value again and is
    split into three.  Reference: float64 on the same bf16 values."""
    from uno_amd import _native
    g = torch.Generator().manual_seed(C1 + C2 + Co + P + 1)
    gy = torch.randn(B, Co, P, generator=g).bfloat16().cuda()
    x1 = torch.randn(B, C1, P, generator=g).bfloat16().cuda()
    x2 = torch.randn(B, C2, P, generator=g).bfloat16().cuda() if C2 else None
    _native.profile_begin(64)
    gw, gb = _native.channel_wgrad2(gy, x1, x2, act_x=act_x)
    torch.cuda.synchronize()
    names = {n for n, _, _ in _native.profile_end()}
    assert "uno::channel_wgrad_split_kernel" in names, names
    xs = [_gelu64(x1.float()) if act_x else x1.double()] + ([x2.double()] if C2 else [])
    assert gw.dtype == torch.float32 and gw.shape == (Co, C1 + C2)
    assert rel(gw, torch.einsum("bop,bip->oi", gy.double(), torch.cat(xs, 1))) < 2e-6
    assert rel(gb, gy.double().sum(dim=(0, 2))) < 2e-6


def test_small_wide_layers_keep_the_vector_kernel():
    """below 100 000 pixels per launch the f32-MFMA vector kernel keeps the layer (the NS-2D roll-out's wide layers: A/B in DESIGN.md)"""
    from uno_amd import _native
    gy, x = torch.randn(32, 192, 1024, device="cuda"), torch.randn(32, 96, 1024, device="cuda")
    _native.profile_begin(16)
    gw, _ = _native.channel_wgrad(gy, x)
    torch.cuda.synchronize()
    names = {n for n, _, _ in _native.profile_end()}
    assert "uno::channel_wgrad_vec_kernel" in names and "uno::channel_wgrad_split_kernel" not in names, names
    assert rel(gw, torch.einsum("bop,bip->oi", gy.double(), x.double())) < 2e-5


def test_wide_wgrad_red_zone():
    """the partial-sum workspace of the split form is sized by the same plan the kernel follows: nothing is written past it"""
    from uno_amd import _native
    B, Ci, Co, P = 2, 96, 130, 50333
    gy, x = torch.randn(B, Co, P, device="cuda"), torch.randn(B, Ci, P, device="cuda")
    nf = _native.channel_wgrad_partial_floats(B, Ci, Co, P)
    raw = torch.full((nf + 2048,), 3.25, device="cuda")
    parts = raw[1024:1024 + nf]
    assert _native.channel_wgrad2(gy, x, None, partials_out=parts) == (None, None)
    torch.cuda.synchronize()
    assert bool((raw[:1024] == 3.25).all()) and bool((raw[1024 + nf:] == 3.25).all())
    gw, gb = _native.channel_wgrad_finish(parts.view(1, -1), Ci, Co, True)
    assert rel(gw, torch.einsum("bop,bip->oi", gy.double(), x.double())) < 2e-6 and rel(gb, gy.double().sum(dim=(0, 2))) < 2e-6


# ---- K8-S on PRE-SPLIT weights (round 5): with uno_channel_mix_ws_bytes() of scratch registered the weights are split once per call
# (channel_mix_wsplit_kernel) into the kernel's LDS images - the binding always provides it, so every K8-S test above runs that form.
# Here the same calls run WITHOUT scratch as well (the weights split by every workgroup, rounds 3-4): same arithmetic on the same
# pairs, so the results must be bit-identical.
def _without_scratch(fn):
    from uno_amd import _native
    init = _native._mix_scratch.__init__

    def none(self, device, *a):
        self.bytes, self.device, self.buf = 0, device, None
    _native._mix_scratch.__init__ = none
    try:
        return fn()
    finally:
        _native._mix_scratch.__init__ = init


@pytest.mark.parametrize("B,C1,C2,Co,P", SPLIT + SPLIT64 + [(2, 128, 0, 128, 5 * 128)])
@pytest.mark.parametrize("bf", [False, True])
def test_presplit_weights_equal_the_per_workgroup_split(B, C1, C2, Co, P, bf):
    from uno_amd import _native
    g = torch.Generator().manual_seed(3 * C1 + C2 + Co + P)
    dt = torch.bfloat16 if bf else torch.float32
    x1 = torch.randn(B, C1, P, generator=g).cuda().to(dt)
    x2 = torch.randn(B, C2, P, generator=g).cuda().to(dt) if C2 else None
    w, b = (torch.randn(Co, C1 + C2, generator=g) / (C1 + C2) ** 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    assert _native.lib().uno_channel_mix_ws_bytes(C1 + C2, Co, P, 1 if bf else 0) == 6 * (C1 + C2) * Co
    calls = [lambda: _native.channel_mix2(x1, x2, w, b), lambda: _native.channel_mix2(x1, x2, w, b, act_in=True)]
    if not C2:
        wt = (torch.randn(C1, Co, generator=g) / C1 ** 0.5).cuda()
        pre = torch.randn(B, Co, P, generator=g).cuda().to(dt)
        calls += [lambda: _native.channel_mix(x1, wt, None, transpose_w=True), lambda: _native.channel_mix(x1, wt, None, transpose_w=True, dgelu_of=pre)]
    for call in calls:
        (got, names) = _launched(call)
        assert names == ["uno::channel_mix_split_kernel"], names
        assert torch.equal(got, _without_scratch(call))


def test_channel_mix_scratch_is_optional_and_sized_by_the_library():
    from uno_amd import _native
    L = _native.lib()
    assert L.uno_channel_mix_ws_bytes(64, 64, 4096, 0) == 0 and L.uno_channel_mix_ws_bytes(128, 100, 4096, 0) == 0
    assert L.uno_channel_mix_ws_bytes(128, 64, 100, 0) == 0 and L.uno_channel_mix_ws_bytes(64, 64, 4096, 1) == 6 * 64 * 64
    x, w = torch.randn(2, 128, 640).cuda(), torch.randn(128, 128).cuda()
    ref = _without_scratch(lambda: _native.channel_mix(x, w, None))
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")               # too small: ignored, not an error
    L.uno_scratch_provide(small.data_ptr(), small.numel())
    try:
        y = torch.empty_like(ref)
        rc = L.uno_channel_mix(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 2, 128, 128, 640, 0, 0, 0, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    finally:
        L.uno_scratch_provide(None, 0)
    assert torch.equal(y, ref)
