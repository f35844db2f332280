import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from uno_amd import _native
B, C, S, m = 16, 64, 421, 20
dev = torch.device("cuda:0")
x = torch.randn(B, C, S, S, device=dev); gy = torch.randn(B, C, S, S, device=dev)
w1 = (0.1 * torch.randn(C, C, m, m, dtype=torch.cfloat)).to(dev); w2 = (0.1 * torch.randn(C, C, m, m, dtype=torch.cfloat)).to(dev)
y, xt = _native.spectral_conv2d_forward(x, w1, w2, S, S)
for _ in range(3): _native.spectral_conv2d_backward(gy, xt, w1, w2, S, S)
torch.cuda.synchronize()
_native.profile_begin(1000)
for _ in range(5):
    _native.spectral_conv2d_forward(x, w1, w2, S, S)
    _native.spectral_conv2d_backward(gy, xt, w1, w2, S, S)
torch.cuda.synchronize()
for name, ms, by in _native.profile_end()[-7:]:
    print(f"{name:42s} {ms*1e3:8.1f} us")
