"""C4 3-D spectral convolution with COLD inputs (the input volumes rotate over more copies than the 256 MB Infinity Cache holds), as a
rocprofv3 --kernel-trace --stats target:  python tools/dev/vol3dcold.py [lib.so|-] [fwd|bwd] [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device("cuda:0")
B, C, dims, modes = 8, 32, (64, 64, 20), (16, 16, 8)
g = torch.Generator().manual_seed(0)
xs = [torch.randn(B, C, *dims, generator=g).to(dev) for _ in range(7)]
ws = [(0.1 * torch.randn(C, C, *modes, dtype=torch.cfloat, generator=g)).to(dev) for _ in range(4)]
y, xt = _native.spectral_conv3d_forward(xs[0], ws, *dims)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    torch.cuda.synchronize()
    e0.record()
    for i in range(calls):
        if which == "fwd":
            _native.spectral_conv3d_forward(xs[i % 7], ws, *dims)
        else:
            _native.spectral_conv3d_backward(xs[i % 7], xt, ws, *dims)
    e1.record()
    torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else '-'} {which}: {e0.elapsed_time(e1) / calls * 1e3:.1f} us per call (cold inputs)")
