import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, H, W, T, m1, m2, m3 = 8, 32, 64, 64, 20, 16, 16, 8
x = torch.randn(B, C, H, W, T, generator=g).to(dev)
ws = [(0.1 * torch.randn(C, C, m1, m2, m3, dtype=torch.cfloat, generator=g)).to(dev) for _ in range(4)]
gy = torch.randn(B, C, H, W, T, generator=g).to(dev)
for _ in range(3):
    y, xt = _native.spectral_conv3d_forward(x, ws, H, W, T)
    _native.spectral_conv3d_backward(gy, xt, ws, H, W, T)
torch.cuda.synchronize()
for name, fn in (("forward", lambda: _native.spectral_conv3d_forward(x, ws, H, W, T)), ("backward", lambda: _native.spectral_conv3d_backward(gy, xt, ws, H, W, T))):
    _native.profile_begin(1000)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    rec = _native.profile_end()
    agg = {}
    order = []
    for n, ms, by in rec:
        if n not in agg:
            agg[n] = [0, 0.0, 0.0]; order.append(n)
        agg[n][0] += 1; agg[n][1] += ms; agg[n][2] += by
    print(name, "total us/call", sum(v[1] for v in agg.values()) / 5 * 1e3)
    for n in order:
        c, ms, by = agg[n]
        print(f"   {n:50s} {c//5:2d}/call  {ms/c*1e3:7.1f} us each  {by/ms/1e6:8.0f} GB/s")
