"""Does the time of the fused fc0 call depend on WHERE its three tensors lie relative to each other?  (python tools/dev/fc0offsets.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
dev = torch.device("cuda:0")
B, Ci, Co, H, W, Hp, Wp = 16, 32, 64, 421, 421, 446, 446
L = _native.lib()
torch.manual_seed(0)
w = (torch.randn(Co, Ci) / 6).to(dev); b = torch.randn(Co).to(dev)
nx, ny, na = B * Ci * H * W, B * Co * H * W, B * Co * Hp * Wp
pool = torch.empty(nx + ny + na + (64 << 20), dtype=torch.float32, device=dev)
pool[:nx].normal_()
st = torch.cuda.current_stream().cuda_stream


def timed(yoff, aoff, n=10):
    x = pool.data_ptr()
    y = pool.data_ptr() + 4 * (nx + yoff)
    a = pool.data_ptr() + 4 * (nx + ny + (16 << 20) + aoff)
    def call():
        rc = L.uno_channel_mix_act_padded(x, w.data_ptr(), b.data_ptr(), y, a, B, Ci, Co, H, W, Hp, Wp, 1, st)
        assert rc == 0
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for yoff, aoff in [(0, 0), (0, 64), (0, 1024), (0, 16384), (0, 1 << 18), (0, 1 << 20), (0, 3 << 20), (64, 0), (1024, 0), (1 << 18, 0), (12345 * 4, 54321 * 4), (0, 0)]:
    print(f"y offset {yoff:9d} floats, act offset {aoff:9d} floats: {timed(yoff, aoff):7.1f} us (call + border clear)", flush=True)
