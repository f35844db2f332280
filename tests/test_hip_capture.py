"""A shape the library has not seen before may first appear INSIDE a hipGraph capture: its operand tables (twiddles of the 2-D
kernels, the 3-D volume kernels' operand tables, the bf16 transforms' split tables) are built on the spot - allocation and upload
under the relaxed capture mode on a stream of their own (uno_common.h: upload_table) - without ending the capture, and the replay
computes what the eager call computes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _capture_then_compare(layer, x, call):
    # no eager warm-up of THIS shape: the capture is the first time the library sees it
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        y = call(layer, x)
    graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        want = call(layer, x)
    assert torch.isfinite(y.float()).all()
    assert torch.equal(y, want)


def test_first_seen_2d_grid_inside_a_capture():
    from uno_amd.integral_operators import SpectralConv2d_Uno
    torch.manual_seed(0)
    layer = SpectralConv2d_Uno(4, 6, 59, 61, 7, 5).cuda()          # 59 / 61: grid sizes no other test uses
    x = torch.randn(2, 4, 67, 71).cuda()                          # ... nor 67 / 71 on the input side
    _capture_then_compare(layer, x, lambda m, v: m(v))


def test_first_seen_3d_grid_inside_a_capture():
    from uno_amd.integral_operators import SpectralConv3d_Uno
    torch.manual_seed(0)
    layer = SpectralConv3d_Uno(4, 4, 22, 26, 14, 4, 5, 3).cuda()
    x = torch.randn(2, 4, 22, 26, 14).cuda()
    _capture_then_compare(layer, x, lambda m, v: m(v))


def test_first_seen_bf16_grid_inside_a_capture():
    from uno_amd.integral_operators import SpectralConv2d_Uno, enable_mixed_precision
    torch.manual_seed(0)
    layer = enable_mixed_precision(SpectralConv2d_Uno(8, 8, 58, 62, 6, 6).cuda())
    x = torch.randn(2, 8, 58, 62).cuda().bfloat16()
    _capture_then_compare(layer, x, lambda m, v: m(v))
