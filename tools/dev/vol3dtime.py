"""Per-kernel times of the 3-D spectral convolution at the layer shapes of Uno3D_T20 (width w, batch 8):
    python tools/dev/vol3dtime.py [lib.so|-] [w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
w = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
B = 8
LAYERS = [  # Ci, Co, in dims, out dims, modes
    (w, 2 * w, (64, 64, 13), (48, 48, 13), (22, 22, 5)),
    (2 * w, 4 * w, (48, 48, 13), (32, 32, 13), (14, 14, 5)),
    (4 * w, 8 * w, (32, 32, 13), (16, 16, 15), (6, 6, 5)),
    (8 * w, 16 * w, (16, 16, 15), (16, 16, 15), (6, 6, 6)),
    (16 * w, 4 * w, (16, 16, 15), (32, 32, 23), (6, 6, 6)),
    (8 * w, 2 * w, (32, 32, 23), (48, 48, 26), (14, 14, 8)),
    (4 * w, 2 * w, (48, 48, 26), (64, 64, 20), (22, 22, 8)),
]
g = torch.Generator().manual_seed(0)
for Ci, Co, din, dout, modes in LAYERS:
    x = torch.randn(B, Ci, *din, generator=g).to(dev)
    ws = [(0.1 * torch.randn(Ci, Co, *modes, dtype=torch.cfloat, generator=g)).to(dev) for _ in range(4)]
    gy = torch.randn(B, Co, *dout, generator=g).to(dev)
    y, xt = _native.spectral_conv3d_forward(x, ws, *dout)
    fwd = lambda: _native.spectral_conv3d_forward(x, ws, *dout)
    bwd = lambda: _native.spectral_conv3d_backward(gy, xt, ws, *din)
    tf, tb = bench._timed(fwd, dev, iters=10, reps=3), bench._timed(bwd, dev, iters=10, reps=3)
    print(f"{Ci}->{Co} {din}->{dout} m{modes}: fwd {tf*1e6:.0f} us bwd {tb*1e6:.0f} us")
    for tag, fn in (("fwd", fwd), ("bwd", bwd)):
        for k, v in bench._kernel_table(fn).items():
            print(f"     {tag} {k} x{v['launches_per_call']:.0f} {v['avg_us']:.1f}")
