"""Channels-last activations at the blocks' boundary (uno_transpose_batched, ABI 8): the reference's blocks accept any strides
(integral_operators.py:187, 233 go through torch.fft / F.conv) and its model files hand conv0 the `permute`d + padded output of a
channels-last nn.Linear (darcy_flow_uno2d.py:104-107).  The product converts such inputs (and channels-last gradients) with its own
tiled transposing copy instead of torch's strided one; results must be bit-identical to those on a contiguous input.  pytest -m gpu"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(2, 32, 446, 446), (3, 5, 37, 41), (1, 96, 64, 33), (2, 64, 1, 7), (2, 130, 9, 9), (1, 2, 1, 1), (4, 7, 3, 5, 6),
                                   (2, 16, 12, 12, 9)])
def test_transposing_copy_both_directions(shape):
    from uno_amd import _native
    g = torch.Generator().manual_seed(0)
    B, C = shape[:2]
    x = torch.randn(B, *shape[2:], C, generator=g).to(dev())            # channels-last memory
    view = x.movedim(-1, 1)                                            # (B, C, *grid) view of it
    assert _native.channels_last_pitch(view) == (C, x[0].numel())
    y = _native.to_channels_first(view)
    assert y.is_contiguous() and torch.equal(y, view.contiguous())
    # the other direction through the C entry point: rows = channels, columns = grid points
    import ctypes as Cc
    P = x[0].numel() // C
    back = torch.empty_like(x)
    rc = _native.lib().uno_transpose_batched(Cc.c_void_p(y.data_ptr()), Cc.c_void_p(back.data_ptr()), B, C, P, P, C * P, C, C * P,
                                             Cc.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and torch.equal(back, x)


def test_transposing_copy_of_channel_slices_and_odd_alignment():
    """a channel slice of a wider channels-last tensor (what torch.cat's backward hands a block: darcy_flow_uno2d.py:125): pitch > C,
    and a slice that starts at an odd channel leaves the rows at 4-byte alignment (scalar load path)"""
    from uno_amd import _native
    g = torch.Generator().manual_seed(1)
    wide = torch.randn(2, 19, 23, 96, generator=g).to(dev()).movedim(-1, 1)     # (2, 96, 19, 23)
    for lo, hi in ((0, 64), (64, 96), (1, 33), (3, 4 + 3), (5, 96)):
        s = wide[:, lo:hi]
        if hi - lo < 2:
            continue
        assert _native.channels_last_pitch(s) is not None
        assert torch.equal(_native.to_channels_first(s), s.contiguous())
    # layouts the fast path must refuse
    assert _native.channels_last_pitch(wide[..., :20]) is None                  # cropped rows: grid points not at one pitch
    assert _native.channels_last_pitch(wide.contiguous()) is None
    assert _native.channels_last_pitch(wide[:, ::2]) is None


def test_transposing_copy_red_zone():
    """ragged tiles must not store outside the output"""
    from uno_amd import _native
    for shape in [(2, 5, 37, 41), (1, 67, 9, 13), (3, 33, 65, 1)]:
        B, C = shape[:2]
        x = torch.randn(B, *shape[2:], C, device=dev()).movedim(-1, 1)
        n = x.numel()
        raw = torch.full((n + 2048,), 7.5, device=dev())
        out = raw[1024:1024 + n].view(x.shape)
        import ctypes as Cc
        P = shape[2] * shape[3]
        rc = _native.lib().uno_transpose_batched(Cc.c_void_p(x.data_ptr()), Cc.c_void_p(out.data_ptr()), B, P, C, C, C * P, P, C * P,
                                                 Cc.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(out, x.contiguous())
        assert bool((raw[:1024] == 7.5).all()) and bool((raw[1024 + n:] == 7.5).all())


def _block(normalize, cin=12, cout=10, d=(30, 28), modes=(5, 6)):
    from uno_amd.integral_operators import OperatorBlock_2D
    torch.manual_seed(3)
    return OperatorBlock_2D(cin, cout, d[0], d[1], modes[0], modes[1], Normalize=normalize).to(dev())


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("geom", [(45, 41, 30, 28), (20, 22, 30, 28), (30, 28, 30, 28)])
def test_block_on_channels_last_input_equals_contiguous_input(geom, normalize):
    H, W, Ho, Wo = geom
    blk = _block(normalize, d=(Ho, Wo))
    g = torch.Generator().manual_seed(5)
    base = torch.randn(2, H, W, 12, generator=g).to(dev())
    gy = torch.randn(2, 10, Ho, Wo, generator=g).to(dev())

    def run(x, gout):
        x = x.detach().requires_grad_(True)
        for p in blk.parameters():
            p.grad = None
        y = blk(x)
        y.backward(gout)
        return y.detach(), x.grad, [p.grad.clone() for p in blk.parameters()]

    y0, gx0, gp0 = run(base.movedim(-1, 1).contiguous(), gy)                                   # channels-first input and gradient
    y1, gx1, gp1 = run(base.movedim(-1, 1), gy.contiguous(memory_format=torch.channels_last))   # both channels-last
    assert y1.is_contiguous() and torch.equal(y0, y1) and torch.equal(gx0, gx1)
    for a, b in zip(gp0, gp1):
        assert torch.equal(a, b)


def test_block_3d_on_channels_last_input():
    from uno_amd.integral_operators import OperatorBlock_3D
    torch.manual_seed(7)
    blk = OperatorBlock_3D(6, 4, 12, 12, 10, 4, 4, 3).to(dev())
    base = torch.randn(2, 16, 16, 8, 6, device=dev())
    outs = []
    for x in (base.movedim(-1, 1).contiguous(), base.movedim(-1, 1)):
        x = x.detach().requires_grad_(True)
        y = blk(x)
        y.square().sum().backward()
        outs.append((y.detach(), x.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


