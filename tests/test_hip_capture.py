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


def test_first_seen_block_with_resampling_inside_a_capture():
    """the Python-side operand tables (banded interpolation operators, the fused up-sampling operands) are uploaded through
    uno_upload_table while the stream is capturing"""
    from uno_amd.integral_operators import OperatorBlock_2D
    torch.manual_seed(0)
    down = OperatorBlock_2D(4, 6, 37, 43, 5, 5).cuda()             # 73 x 83 -> 37 x 43
    up = OperatorBlock_2D(6, 4, 73, 83, 5, 5).cuda()               # ... and back
    x = torch.randn(2, 4, 73, 83).cuda()
    _capture_then_compare((down, up), x, lambda m, v: m[1](m[0](v, 37, 43), 73, 83))


def test_first_seen_3d_block_inside_a_capture():
    from uno_amd.integral_operators import OperatorBlock_3D
    torch.manual_seed(0)
    blk = OperatorBlock_3D(4, 4, 18, 22, 12, 3, 3, 3).cuda()       # 26 x 30 x 16 -> 18 x 22 x 12 (kept-index tables of the 3-D resampling)
    x = torch.randn(2, 4, 26, 30, 16).cuda()
    _capture_then_compare(blk, x, lambda m, v: m(v, 18, 22, 12))
