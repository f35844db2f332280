"""Standalone spectral blocks for rocprofv3: N forward + N backward calls of
    c2: SpectralConv2d(64,64,421,421,20,20), batch 16      (BASELINE.json's roofline block; SURVEY 8(d): fwd 1478.2 MB, bwd 1504.4 MB)
    c4: SpectralConv3d(32,32,64,64,20,16,16,8), batch 8
usage: python tools/block_prof.py c2|c4 [N]      (run under rocprofv3 --kernel-trace --stats / --pmc ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
if which == "c2":
    B, C, S, m = 16, 64, 421, 20
    x = torch.randn(B, C, S, S, generator=g).to(dev)
    sc = (1 / (2 * C)) ** 0.5
    w1 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
    w2 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
    gy = torch.randn(B, C, S, S, generator=g).to(dev)
    y, xt = _native.spectral_conv2d_forward(x, w1, w2, S, S)
    for _ in range(N):
        _native.spectral_conv2d_forward(x, w1, w2, S, S)
    torch.cuda.synchronize()
    for _ in range(N):
        _native.spectral_conv2d_backward(gy, xt, w1, w2, S, S)
else:
    B, C, H, W, T, m1, m2, m3 = 8, 32, 64, 64, 20, 16, 16, 8
    x = torch.randn(B, C, H, W, T, generator=g).to(dev)
    sc = (1 / (2 * C)) ** 0.5
    ws = [(sc * torch.randn(C, C, m1, m2, m3, dtype=torch.cfloat, generator=g)).to(dev) for _ in range(4)]
    gy = torch.randn(B, C, H, W, T, generator=g).to(dev)
    y, xt = _native.spectral_conv3d_forward(x, ws, H, W, T)
    for _ in range(N):
        _native.spectral_conv3d_forward(x, ws, H, W, T)
    torch.cuda.synchronize()
    for _ in range(N):
        _native.spectral_conv3d_backward(gy, xt, ws, H, W, T)
torch.cuda.synchronize()
print("done", which, N)
