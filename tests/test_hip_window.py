"""Windowed channel-mix calls (uno_channel_mix2_win / uno_channel_wgrad2_win / uno_gelu_project_backward_win, ABI 10): the last
two layers of the Darcy model on the S x S domain INSIDE the padded tensors (reference darcy_flow_uno2d.py:125-131 crops the
padding, then fc1 - GELU - fc2).  Every windowed call is compared with the dense call on a contiguous copy of the window, in
float64 where the dense call's own tests do; elements outside the window must be left exactly as they were."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# rows, cols (multiple of 4, >= 260), pitch, plane rows of the padded tensor
WINDOWS = [
    (9, 264, 264, 9),            # the window IS the plane
    (7, 260, 301, 11),           # odd pitch: rows start 4-byte aligned only
    (37, 424, 446, 40),          # the Darcy geometry (421 -> 424 of 446), a few rows
    (5, 512, 515, 5),            # full height, three spare columns
]


def rel(a, b):
    d = (a.double() - b.double()).norm().item()
    n = b.double().norm().item()
    return d / n if n > 0 else d


def _planes(B, C, H, pitch, g, fill=None):
    t = torch.randn(B, C, H * pitch, generator=g)
    if fill is not None:
        t.fill_(fill)
    return t.cuda()


def _crop(t, rows, cols, pitch):
    """(B, C, plane) -> contiguous (B, C, rows * cols)"""
    B, C = t.shape[:2]
    return t.view(B, C, -1, pitch)[:, :, :rows, :cols].reshape(B, C, rows * cols).contiguous()


def _outside_untouched(t, before, rows, cols, pitch):
    v, b = t.view(*t.shape[:2], -1, pitch), before.view(*t.shape[:2], -1, pitch)
    return torch.equal(v[:, :, rows:], b[:, :, rows:]) and torch.equal(v[:, :, :rows, cols:], b[:, :, :rows, cols:])


@pytest.mark.parametrize("rows,cols,pitch,H", WINDOWS)
@pytest.mark.parametrize("C1,C2,Co,act_in", [(64, 64, 64, True), (32, 0, 40, False), (128, 0, 128, False), (64, 64, 64, False), (16, 16, 24, True)])
def test_forward_window_equals_dense_on_the_crop(rows, cols, pitch, H, C1, C2, Co, act_in):
    from uno_amd import _native
    g = torch.Generator().manual_seed(rows + cols + C1 + Co)
    B = 2
    x1 = _planes(B, C1, H, pitch, g)
    x2 = _planes(B, C2, H, pitch, g) if C2 else None
    w, b = (torch.randn(Co, C1 + C2, generator=g) / (C1 + C2) ** 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    out = torch.full((B, Co, H * pitch), 7.25).cuda()
    before = out.clone()
    _native.channel_mix2(x1, x2, w, b, act_in=act_in, out=out, window=(rows, cols, pitch))
    ref = _native.channel_mix2(_crop(x1, rows, cols, pitch), _crop(x2, rows, cols, pitch) if C2 else None, w, b, act_in=act_in)
    assert rel(_crop(out, rows, cols, pitch), ref) < 1e-6
    assert _outside_untouched(out, before, rows, cols, pitch)


@pytest.mark.parametrize("rows,cols,pitch,H", WINDOWS)
def test_fused_projection_window(rows, cols, pitch, H):
    """the fc1 - GELU - fc2 kernel of the model (two sources, GELU on the first, 64 channels, one projected output)"""
    from uno_amd import _native
    g = torch.Generator().manual_seed(rows * 3 + cols)
    B, C1, C2, Co = 2, 64, 64, 64
    x1, x2 = _planes(B, C1, H, pitch, g), _planes(B, C2, H, pitch, g)
    w, b = (torch.randn(Co, C1 + C2, generator=g) / 11).cuda(), torch.randn(Co, generator=g).cuda()
    w2, b2 = torch.randn(Co, generator=g).cuda(), torch.randn(1, generator=g).cuda()
    y, proj = _native.channel_mix2(x1, x2, w, b, act_in=True, project=(w2, b2), window=(rows, cols, pitch))
    yr, pr = _native.channel_mix2(_crop(x1, rows, cols, pitch), _crop(x2, rows, cols, pitch), w, b, act_in=True, project=(w2, b2))
    assert y.shape == (B, Co, H * pitch) and proj.shape == (B, H * pitch)
    assert rel(_crop(y, rows, cols, pitch), yr) < 1e-6
    assert rel(_crop(proj.view(B, 1, -1), rows, cols, pitch), pr.view(B, 1, -1)) < 1e-6


@pytest.mark.parametrize("rows,cols,pitch,H", WINDOWS)
@pytest.mark.parametrize("Ci,Co,dgelu,acc", [(64, 64, True, 0), (64, 64, False, 1), (64, 64, True, 2), (64, 32, False, 0), (128, 128, True, 1), (128, 64, False, 0)])
def test_input_gradient_window(rows, cols, pitch, H, Ci, Co, dgelu, acc):
    """transposed calls: fresh output, accumulating output, gelu' on the product (acc 0 / 1) and on the completed sum (acc 2)"""
    from uno_amd import _native
    g = torch.Generator().manual_seed(rows + Ci + Co + acc)
    B = 2
    gy = _planes(B, Ci, H, pitch, g)
    w = (torch.randn(Ci, Co, generator=g) / Ci ** 0.5).cuda()
    pre = _planes(B, Co, H, pitch, g) if dgelu else None
    out = _planes(B, Co, H, pitch, g)
    before = out.clone()
    win = (rows, cols, pitch)
    if acc:
        _native.channel_mix(gy, w, None, transpose_w=True, out=out, dgelu_of=pre, dgelu_total=acc == 2, window=win)
        ref = _crop(before, rows, cols, pitch)
        _native.channel_mix(_crop(gy, *win), w, None, transpose_w=True, out=ref, dgelu_of=_crop(pre, *win) if dgelu else None, dgelu_total=acc == 2)
    else:
        _native.channel_mix2(gy, None, w, None, transpose_w=True, out=out, dgelu_of=pre, window=win)
        ref = _native.channel_mix(_crop(gy, *win), w, None, transpose_w=True, dgelu_of=_crop(pre, *win) if dgelu else None)
    assert rel(_crop(out, *win), ref) < 1e-6
    assert _outside_untouched(out, before, *win)


@pytest.mark.parametrize("rows,cols,pitch,H", WINDOWS)
@pytest.mark.parametrize("C1,C2,Co,act_x", [(64, 64, 64, True), (64, 0, 64, False), (128, 128, 128, False), (64, 64, 32, False), (128, 0, 256, True)])
def test_weight_gradient_window(rows, cols, pitch, H, C1, C2, Co, act_x):
    from uno_amd import _native
    g = torch.Generator().manual_seed(rows + C1 + Co)
    B = 3
    win = (rows, cols, pitch)
    gy, x1 = _planes(B, Co, H, pitch, g), _planes(B, C1, H, pitch, g)
    x2 = _planes(B, C2, H, pitch, g) if C2 else None
    gw, gb = _native.channel_wgrad2(gy, x1, x2, need_bias=True, act_x=act_x, window=win)
    x1c = _crop(x1, *win).double()
    if act_x:
        x1c = torch.nn.functional.gelu(x1c)
    xc = torch.cat([x1c] + ([_crop(x2, *win).double()] if C2 else []), 1)
    gyc = _crop(gy, *win).double()
    assert rel(gw, torch.einsum("bop,bip->oi", gyc, xc)) < 2e-5
    assert rel(gb, gyc.sum((0, 2))) < 2e-5


@pytest.mark.parametrize("rows,cols,pitch,H", WINDOWS)
def test_gelu_project_backward_window(rows, cols, pitch, H):
    from uno_amd import _native
    g = torch.Generator().manual_seed(rows + pitch)
    B, Cc = 2, 64
    win = (rows, cols, pitch)
    pre, gout = _planes(B, Cc, H, pitch, g), _planes(B, 1, H, pitch, g).view(B, -1)
    w = torch.randn(Cc, generator=g).cuda()
    gpre, gw, gb = _native.gelu_project_backward(pre, w, gout, need_bias=True, window=win)
    rp, rw, rb = _native.gelu_project_backward(_crop(pre, *win), w, _crop(gout.view(B, 1, -1), *win).view(B, -1), need_bias=True)
    assert rel(_crop(gpre, *win), rp) < 1e-6 and rel(gw, rw) < 2e-5 and rel(gb, rb) < 2e-5


def test_cat_project_with_crop_matches_the_uncropped_layer():
    """channel_mix_cat_project(crop=): output and every gradient of the cropped result agree with the call that computes the whole
    padded grid and crops afterwards (what the model did before ABI 10), the input gradients are exactly zero outside the domain."""
    from uno_amd.integral_operators import channel_mix_cat_project
    torch.manual_seed(11)
    B, C1, C2, Co, H, W, S1, S2 = 2, 64, 64, 64, 40, 300, 33, 277
    base = [torch.randn(B, C1, H, W), torch.randn(B, C2, H, W), torch.randn(Co, C1 + C2) / 11, torch.randn(Co), torch.randn(1, Co), torch.randn(1)]
    gout = torch.randn(B, 1, S1, S2).cuda()
    res = []
    for crop in ((S1, S2), None):
        t = [v.clone().cuda().requires_grad_(True) for v in base]
        out = channel_mix_cat_project(t[:2], t[2], t[3], t[4], t[5], gelu_first=True, crop=crop)[:, :, :S1, :S2]
        out.backward(gout)
        res.append((out.detach(), [v.grad for v in t]))
    (o1, g1), (o0, g0) = res
    assert rel(o1, o0) < 1e-6
    for a, b, tol in zip(g1, g0, (2e-6, 2e-6, 2e-5, 2e-5, 2e-5, 2e-5)):
        assert rel(a, b) < tol
    for gx in g1[:2]:
        assert float(gx[:, :, S1:].abs().max()) == 0.0 and float(gx[:, :, :, S2:].abs().max()) == 0.0


def test_cropped_projection_deferring_into_a_join_whose_tensor_is_a_fused_gelu_activation():
    """out_join + defer_grad + crop together (advisor finding, round 5): the tensor `a` is the activation of a block that leaves its
    pre-activation sum to the join; its first consumer (a block, the owner) completes the joined gradient; its second consumer is the
    cropped fc1 - GELU - fc2, whose deferred contribution reaches the WINDOW only.  gelu'(pre) must reach the whole plane (the border
    holds the owner's own gradient): every gradient against the same graph without joins and without the crop."""
    import uno_amd.integral_operators as io
    torch.manual_seed(13)
    B, H, W, S1, S2 = 2, 20, 300, 17, 277
    P = io.OperatorBlock_2D(8, 64, H, W, 4, 4).cuda()
    Q = io.OperatorBlock_2D(64, 64, H, W, 4, 4).cuda()
    w = (torch.randn(64, 128) / 11).cuda().requires_grad_(True)
    b = torch.randn(64).cuda().requires_grad_(True)
    w2 = torch.randn(1, 64).cuda().requires_grad_(True)
    b2 = torch.randn(1).cuda().requires_grad_(True)
    x0 = torch.randn(B, 8, H, W).cuda()
    gout = torch.randn(B, 1, S1, S2).cuda()
    leaves = list(P.parameters()) + list(Q.parameters()) + [w, b, w2, b2]

    def run(joined):
        for p in leaves:
            p.grad = None
        x = x0.clone().requires_grad_(True)
        J = io.GradJoin() if joined else None
        a = P(x, H, W, out_join=J) if joined else P(x, H, W)
        q = Q(a, H, W, join=J) if joined else Q(a, H, W)
        out = io.channel_mix_cat_project([q, a], w, b, w2, b2, gelu_first=False, defer_grad=J, crop=(S1, S2) if joined else None)
        (out[:, :, :S1, :S2] * gout).sum().backward()
        return [x.grad.clone()] + [p.grad.clone() for p in leaves]
    got, ref = run(True), run(False)
    for a_, b_ in zip(got, ref):
        assert rel(a_, b_) < 2e-5, rel(a_, b_)


def test_window_argument_errors():
    from uno_amd import _native
    x = torch.randn(1, 64, 20 * 300).cuda()
    w = torch.randn(64, 64).cuda()
    for win in ((20, 258, 300), (20, 262, 300), (20, 304, 300), (21, 260, 300)):
        with pytest.raises(RuntimeError):
            _native.channel_mix2(x, None, w, None, window=win)
    with pytest.raises(RuntimeError):
        _native.channel_mix2(x.bfloat16(), None, w, None, window=(20, 260, 300))
    with pytest.raises(RuntimeError):                                           # few input channels: dense only
        _native.channel_wgrad2(torch.randn(1, 8, 20 * 300).cuda(), torch.randn(1, 3, 20 * 300).cuda(), None, window=(20, 260, 300))


@pytest.mark.parametrize("B,Ci,Co,H,W,ph,pw", [(2, 32, 64, 30, 261, 3, 7), (1, 32, 64, 421, 421, 25, 25), (2, 16, 40, 9, 300, 0, 5), (2, 32, 64, 5, 264, 2, 0)])
def test_lift_with_padded_activation(B, Ci, Co, H, W, ph, pw):
    """uno_channel_mix_act_padded: y = W gelu(x) + b kept, act = zero-pad(gelu(y)) from the same kernel (odd widths: a lane's four
    pixels straddle row ends; partial last pixel tile; partial channel tile)"""
    from uno_amd import _native
    g = torch.Generator().manual_seed(H + W + Co)
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    w, b = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    y, act = _native.channel_mix_act_padded(x, w, b, H + ph, W + pw, act_in=True)
    yr = torch.matmul(w.double(), torch.nn.functional.gelu(x.double()).view(B, Ci, -1)).view(B, Co, H, W) + b.double().view(1, -1, 1, 1)
    ar = torch.nn.functional.pad(torch.nn.functional.gelu(yr), [0, pw, 0, ph])
    assert rel(y, yr) < 2e-6 and act.shape == ar.shape and rel(act, ar) < 2e-6
    assert float(act[:, :, H:].abs().max() if ph else 0) == 0.0 and float(act[:, :, :, W:].abs().max() if pw else 0) == 0.0
    # without the pre-activation result: the same activation, and the backward kernel that recomputes it
    none, act2 = _native.channel_mix_act_padded(x, w, b, H + ph, W + pw, act_in=True, keep_y=False)
    assert none is None and torch.equal(act2, act)
    gp = torch.randn(B, Co, H + ph, W + pw, generator=g).cuda()
    gz = _native.channel_mix_dgelu_padded(x, w, b, gp, act_in=True)
    yd = yr.clone().requires_grad_(True)
    torch.nn.functional.gelu(yd).backward(gp[:, :, :H, :W].double())
    assert gz.shape == (B, Co, H, W) and rel(gz, yd.grad) < 3e-6


def test_gelu_channel_mix_pad_autograd_matches_the_two_step_form():
    from uno_amd.integral_operators import gelu_channel_mix, gelu_channel_mix_pad, gelu_pad2d
    torch.manual_seed(5)
    B, Ci, Co, H, W, pad = 2, 32, 64, 37, 283, 6
    base = [torch.randn(B, Ci, H, W), torch.randn(Co, Ci) / 6, torch.randn(Co)]
    gout = torch.randn(B, Co, H + pad, W + pad).cuda()
    res = []
    for fused in (True, False):
        t = [v.clone().cuda().requires_grad_(True) for v in base]
        out = gelu_channel_mix_pad(t[0], t[1], t[2], pad, pad) if fused else gelu_pad2d(gelu_channel_mix(t[0], t[1], t[2]), pad, pad)
        out.backward(gout)
        res.append((out.detach(), [v.grad for v in t]))
    (o1, g1), (o0, g0) = res
    assert rel(o1, o0) < 1e-6
    for a, b, tol in zip(g1, g0, (2e-6, 2e-5, 2e-5)):
        assert rel(a, b) < tol


def test_clear_border():
    from uno_amd import _native
    t = torch.randn(3, 5, 23, 31).cuda()
    ref = t.clone()
    ref[..., 17:, :] = 0
    ref[..., :, 29:] = 0
    assert torch.equal(_native.clear_border(t, 17, 29), ref)
    t2 = torch.randn(2, 7, 9).cuda()
    assert torch.equal(_native.clear_border(t2.clone(), 7, 9), t2)


@pytest.mark.parametrize("B,Cin,Cm,Co,H,W,ph,pw,bias", [(2, 3, 32, 64, 37, 283, 6, 6, True), (1, 3, 32, 64, 421, 421, 25, 25, True), (2, 1, 16, 24, 9, 300, 0, 5, True),
                                                       (2, 2, 32, 40, 5, 264, 2, 0, False), (3, 3, 16, 128, 4, 260, 1, 1, True),
                                                       # K15 (32 / 64 channels): no padding at all (clamped loads at the planes' ends), a width that is a
                                                       # multiple of 4, one and two real channels, no biases, many tiles per workgroup and a ragged last one
                                                       (2, 3, 32, 64, 7, 261, 0, 0, True), (1, 1, 32, 64, 3, 264, 0, 3, False), (2, 2, 32, 64, 41, 300, 5, 0, True),
                                                       (1, 3, 32, 64, 130, 263, 1, 2, True)])
def test_whole_lift_without_stored_intermediates(B, Cin, Cm, Co, H, W, ph, pw, bias):
    """uno_lift_forward / uno_lift_backward against float64 torch: F.pad(gelu(fc0(gelu(fc_n1(x))))) and the four parameter gradients, with
    the first layer's output virtual in every kernel (interior and edge pixel tiles, partial channel tiles, 1 - 3 real channels)"""
    from uno_amd import _native
    gen = torch.Generator().manual_seed(B + Cin + Cm + Co + H + W)
    x = torch.randn(B, Cin, H, W, generator=gen).cuda()
    w1, w0 = torch.randn(Cm, Cin, generator=gen).cuda(), (torch.randn(Co, Cm, generator=gen) / Cm ** 0.5).cuda()
    b1, b0 = (torch.randn(Cm, generator=gen).cuda(), torch.randn(Co, generator=gen).cuda()) if bias else (None, None)
    act = _native.lift_forward(x, w1, b1, w0, b0, H + ph, W + pw)
    d = [t.double().requires_grad_(True) if t is not None else None for t in (w1, b1, w0, b0)]
    F = torch.nn.functional
    h = torch.einsum("mk,bkhw->bmhw", d[0], x.double()) + (d[1].view(1, -1, 1, 1) if bias else 0)
    z = torch.einsum("om,bmhw->bohw", d[2], F.gelu(h)) + (d[3].view(1, -1, 1, 1) if bias else 0)
    ref = F.pad(F.gelu(z), [0, pw, 0, ph])
    assert act.shape == ref.shape and rel(act, ref.detach()) < 3e-6
    assert float(act[:, :, H:].abs().max() if ph else 0) == 0.0 and float(act[:, :, :, W:].abs().max() if pw else 0) == 0.0
    g = torch.randn(act.shape, generator=gen).cuda()
    ref.backward(g.double())
    got = _native.lift_backward(x, w1, b1, w0, b0, g)
    for a, r in zip(got, d):
        assert (a is None) == (r is None)
        if a is not None:
            assert rel(a, r.grad) < 3e-5


@pytest.mark.parametrize("B,Cin,H,W,ph,pw", [(2, 3, 37, 283, 6, 6), (1, 3, 421, 421, 25, 25), (2, 2, 41, 300, 5, 0), (1, 1, 130, 263, 1, 2)])
def test_lift_backward_adds_a_second_gradient_as_it_reads(B, Cin, H, W, ph, pw):
    """uno_lift_backward2: the gradient of the lift's output arrives as TWO tensors (its two consumers, reference
    darcy_flow_uno2d.py:108, :127); the second is valid on the domain only - NaN outside it - and the result equals the call on their sum."""
    from uno_amd import _native
    gen = torch.Generator().manual_seed(B + Cin + H + W)
    x = torch.randn(B, Cin, H, W, generator=gen).cuda()
    w1, w0 = torch.randn(32, Cin, generator=gen).cuda(), (torch.randn(64, 32, generator=gen) / 32 ** 0.5).cuda()
    b1, b0 = torch.randn(32, generator=gen).cuda(), torch.randn(64, generator=gen).cuda()
    assert _native.lift_backward_takes_second(x, w1, w0, H + ph, W + pw)
    g = torch.randn(B, 64, H + ph, W + pw, generator=gen).cuda()
    g2 = torch.full((B, 64, H + ph, W + pw), float("nan")).cuda()
    g2[:, :, :H, :W] = torch.randn(B, 64, H, W, generator=gen).cuda()
    both = g.clone()
    both[:, :, :H, :W] += g2[:, :, :H, :W]
    got = _native.lift_backward(x, w1, b1, w0, b0, g, g2)
    ref = _native.lift_backward(x, w1, b1, w0, b0, both)
    for a, r in zip(got, ref):
        assert torch.isfinite(a).all()
        assert rel(a, r) < 2e-6, rel(a, r)


def test_two_destination_input_gradient_on_a_window():
    """both input gradients of a two-source layer from one pass over gy on a window (what fc1's backward hands to the lift's join):
    each equals the one-destination windowed call"""
    from uno_amd import _native
    gen = torch.Generator().manual_seed(21)
    B, C1, C2, Co, H, pitch, rows, cols = 2, 64, 64, 64, 21, 300, 19, 280
    win = (rows, cols, pitch)
    gy = torch.randn(B, Co, H * pitch, generator=gen).cuda()
    w = (torch.randn(Co, C1 + C2, generator=gen) / 8).cuda()
    dg = torch.randn(B, C1, H * pitch, generator=gen).cuda()
    g1, g2 = _native.channel_mix2(gy, None, w, None, transpose_w=True, split_out=C1, dgelu_of=dg, window=win)
    r1 = _native.channel_mix(gy, w[:, :C1].contiguous(), None, transpose_w=True, dgelu_of=dg, window=win)
    r2 = _native.channel_mix(gy, w[:, C1:].contiguous(), None, transpose_w=True, window=win)
    for a, r in ((g1, r1), (g2, r2)):
        a4, r4 = a.view(B, -1, H, pitch)[..., :rows, :cols], r.view(B, -1, H, pitch)[..., :rows, :cols]
        assert rel(a4, r4) < 2e-6


def test_lift_gelu_pad_autograd_matches_layer_by_layer():
    from uno_amd.integral_operators import channel_mix, gelu_channel_mix, gelu_pad2d, lift_gelu_pad
    torch.manual_seed(9)
    B, H, W, pad = 2, 33, 270, 7
    x = torch.randn(B, 3, H, W).cuda()
    gout = torch.randn(B, 64, H + pad, W + pad).cuda()
    res = []
    for fused in (True, False):
        torch.manual_seed(1)
        f1, f0 = torch.nn.Linear(3, 32).cuda(), torch.nn.Linear(32, 64).cuda()
        out = lift_gelu_pad(x, f1, f0, pad, pad) if fused else gelu_pad2d(gelu_channel_mix(channel_mix(x, f1.weight, f1.bias), f0.weight, f0.bias), pad, pad)
        out.backward(gout)
        res.append((out.detach(), [p.grad for p in (f1.weight, f1.bias, f0.weight, f0.bias)]))
    (o1, g1), (o0, g0) = res
    assert rel(o1, o0) < 2e-6
    for a, b in zip(g1, g0):
        assert rel(a, b) < 3e-5
