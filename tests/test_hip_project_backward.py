"""uno_project_backward (ABI 12): the backward pass of the models' last two layers - `fc2(F.gelu(fc1(torch.cat([x_c5, x_fc0], 1))))`,
reference darcy_flow_uno2d.py:125-131 - with the gradient at fc1's output formed inside the input-gradient and the weight-gradient
kernels instead of written by uno_gelu_project_backward and read back twice.  Every gradient against float64 autograd of the same
expression; the autograd function with the fused form switched on and off; shapes outside the kernels' range keep the three calls."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    d = (a.double() - b.double()).norm().item()
    n = b.double().norm().item()
    return d / n if n > 0 else d


def _crop(t, win):
    if win is None:
        return t
    rows, cols, pitch = win
    return t.view(*t.shape[:-1], -1, pitch)[..., :rows, :cols].reshape(*t.shape[:-1], rows * cols).contiguous()


def _reference(x1, x2, w, b, w2, b2, gout, act_in):
    """float64 autograd of the two layers on dense (cropped) tensors -> g1, g2, gw, gb, gw2, gb2"""
    t = [v.double().detach().requires_grad_(True) if v is not None else None for v in (x1, x2, w, b, w2, b2)]
    a = F.gelu(t[0]) if act_in else t[0]
    if t[1] is not None:
        a = torch.cat([a, t[1]], 1)
    pre = torch.einsum("oi,bip->bop", t[2], a) + t[3][None, :, None]
    out = torch.einsum("o,bop->bp", t[4], F.gelu(pre)) + t[5]
    (out * gout.double()).sum().backward()
    return [None if v is None else v.grad for v in t]


# B, C1, C2, Co, (rows, cols, pitch, plane rows) or dense pixel count, act_in
CASES = [
    (3, 64, 64, 64, (130, 264, 270, 133), True),       # the Darcy layer on a window; last pixel tile partial (34 320 = 268 x 128 + 16), tail half chunk
    (3, 64, 64, 96, (130, 264, 270, 133), False),      # 128-row weight tiles with 32 rows missing
    (3, 64, 64, 48, 34000, True),                      # dense, 64-row tile with 16 rows missing
    (3, 128, 0, 64, (130, 264, 270, 133), False),      # one source
    (2, 128, 128, 64, 50000, True),                    # 256 input channels: two 128-channel gradient tiles
    (1, 64, 64, 112, (421, 424, 446, 446), True),      # one image of the headline geometry
]


@pytest.mark.parametrize("B,C1,C2,Co,geo,act_in", CASES)
def test_project_backward_against_float64(B, C1, C2, Co, geo, act_in):
    from uno_amd import _native
    g = torch.Generator().manual_seed(B * 1000 + Co + C2)
    if isinstance(geo, tuple):
        rows, cols, pitch, H = geo
        win, P = (rows, cols, pitch), H * pitch
    else:
        win, P = None, geo
    Ci = C1 + C2
    x1 = torch.randn(B, C1, P, generator=g).cuda()
    x2 = torch.randn(B, C2, P, generator=g).cuda() if C2 else None
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda()
    b, w2, b2 = torch.randn(Co, generator=g).cuda(), torch.randn(Co, generator=g).cuda(), torch.randn(1, generator=g).cuda()
    gout = torch.randn(B, P, generator=g).cuda()
    assert _native.project_backward_applies(B, C1, Ci, Co, P, win)
    # fc1's output as the forward pass keeps it
    a = F.gelu(x1) if act_in else x1
    pre = torch.einsum("oi,bip->bop", w, a if x2 is None else torch.cat([a, x2], 1)) + b[None, :, None]
    pre = pre.contiguous()
    g1, g2, gw, gb, gw2, gb2 = _native.project_backward(x1, x2, w, pre, w2, gout, act_in=act_in, window=win)
    r = _reference(_crop(x1, win), None if x2 is None else _crop(x2, win), w, b, w2, b2, _crop(gout, win), act_in)
    assert rel(_crop(g1, win), r[0]) < 5e-6
    if x2 is not None:
        assert rel(_crop(g2, win), r[1]) < 5e-6
    assert rel(gw, r[2]) < 2e-5 and rel(gb, r[3]) < 2e-5 and rel(gw2, r[4]) < 2e-5 and rel(gb2, r[5]) < 2e-5
    # into given buffers, accumulating
    ow, ob = torch.ones(Co, Ci).cuda(), torch.ones(Co).cuda()
    _native.project_backward(x1, x2, w, pre, w2, gout, act_in=act_in, window=win, out_w=ow, out_b=ob, accumulate=True)
    assert rel(ow - 1, gw) < 1e-5 and rel(ob - 1, gb) < 1e-5
    # run to run: bit for bit (ten launches: a packed accumulation of the projection's weight gradient once lost terms in one launch of four)
    for _ in range(10):
        again = _native.project_backward(x1, x2, w, pre, w2, gout, act_in=act_in, window=win)
        for u, v in zip((g1, gw, gb, gw2, gb2), (again[0], again[2], again[3], again[4], again[5])):
            assert torch.equal(_crop(u, win) if u.dim() == 3 else u, _crop(v, win) if v.dim() == 3 else v)


def test_window_leaves_the_rest_of_the_planes_alone():
    """nothing outside the window is written: the gradient planes keep what they held (the caller clears the border)"""
    import ctypes as C
    from uno_amd import _native
    g = torch.Generator().manual_seed(5)
    B, C1, C2, Co, rows, cols, pitch, H = 3, 64, 64, 64, 130, 264, 270, 133
    P, Ci = H * pitch, 128
    x1, x2 = torch.randn(B, C1, P, generator=g).cuda(), torch.randn(B, C2, P, generator=g).cuda()
    w = (torch.randn(Co, Ci, generator=g) / 11).cuda()
    pre, w2, gout = torch.randn(B, Co, P, generator=g).cuda(), torch.randn(Co, generator=g).cuda(), torch.randn(B, P, generator=g).cuda()
    g1, g2 = torch.full((B, C1, P), 7.0).cuda(), torch.full((B, C2, P), -3.0).cuda()
    gw, gb, gw2, gb2 = torch.empty(Co, Ci).cuda(), torch.empty(Co).cuda(), torch.empty(Co).cuda(), torch.empty(1).cuda()
    L = _native.lib()
    ws = torch.empty(L.uno_project_backward_ws_bytes(B, Ci, Co, rows * cols), dtype=torch.uint8).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = L.uno_project_backward(p(x1), p(x2), C1, p(w), p(pre), p(w2), p(gout), p(g1), p(g2), p(gw), p(gb), p(gw2), p(gb2), p(ws), B, Ci, Co,
                                rows, cols, pitch, P, 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    for t, v in ((g1, 7.0), (g2, -3.0)):
        t4 = t.view(B, -1, H, pitch)
        assert bool((t4[:, :, rows:] == v).all()) and bool((t4[:, :, :rows, cols:] == v).all())
        assert not bool((t4[:, :, :rows, :cols] == v).any())


@pytest.mark.parametrize("crop", [(290, 277), None])
def test_autograd_function_fused_equals_three_calls(crop):
    """channel_mix_cat_project's backward pass with PROJECT_BACKWARD_FUSED on and off: every gradient agrees (and the fused form ran)"""
    import uno_amd.integral_operators as io
    from uno_amd import _native
    torch.manual_seed(3)
    B, C1, C2, Co, H, W = 2, 64, 64, 64, 300, 300
    base = [torch.randn(B, C1, H, W), torch.randn(B, C2, H, W), torch.randn(Co, C1 + C2) / 11, torch.randn(Co), torch.randn(1, Co), torch.randn(1)]
    S1, S2 = crop if crop else (H, W)
    gout = torch.randn(B, 1, S1, S2).cuda()
    res, names = [], []
    for fused in (True, False):
        old = io.PROJECT_BACKWARD_FUSED
        io.PROJECT_BACKWARD_FUSED = fused
        try:
            t = [v.clone().cuda().requires_grad_(True) for v in base]
            out = io.channel_mix_cat_project(t[:2], t[2], t[3], t[4], t[5], gelu_first=True, crop=crop)[:, :, :S1, :S2]
            _native.profile_begin(64)
            out.backward(gout)
            torch.cuda.synchronize()
            names.append([r[0] for r in _native.profile_end()])
            res.append([v.grad for v in t])
        finally:
            io.PROJECT_BACKWARD_FUSED = old
    assert not any("gelu_project_bwd" in n for n in names[0]) and any("gelu_project_bwd" in n for n in names[1])
    for a, b, tol in zip(res[0], res[1], (5e-6, 5e-6, 2e-5, 2e-5, 2e-5, 2e-5)):
        assert rel(a, b) < tol
    if crop:
        for gx in res[0][:2]:
            assert float(gx[:, :, S1:].abs().max()) == 0.0 and float(gx[:, :, :, S2:].abs().max()) == 0.0


def test_shapes_outside_the_range_are_refused():
    from uno_amd import _native
    assert not _native.project_backward_applies(2, 64, 128, 64, 40 * 300, None)           # fewer than 100 000 pixels: the vector weight-gradient kernel
    assert not _native.project_backward_applies(3, 64, 128, 128, 50000, None)             # 128 channels between the layers
    assert not _native.project_backward_applies(3, 32, 128, 64, 50000, None)              # sources split at 32
    assert not _native.project_backward_applies(3, 64, 128, 64, 50001, None)              # planes of odd length
    x1 = torch.randn(2, 64, 12000).cuda()
    with pytest.raises(RuntimeError, match="uno_project_backward_applies"):
        _native.project_backward(x1, x1.clone(), torch.randn(64, 128).cuda(), torch.randn(2, 64, 12000).cuda(), torch.randn(64).cuda(),
                                 torch.randn(2, 12000).cuda())
