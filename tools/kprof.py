"""Launch each spectral kernel a few times at the C2 block size (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
B, C, S, m = 16, 64, 421, 20
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, S, S, generator=g).to(dev)
w1 = (0.1 * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
w2 = (0.1 * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
for _ in range(n):
    X = _native.dft2d_forward(x, m, m, scale=1.0 / (S * S))
    O = _native.mode_mix(X, [w1, w2], 0)
    y = _native.dft2d_inverse(O, S, S)
    gw = _native.mode_wgrad(X, O, w1.shape, 2)
torch.cuda.synchronize()
print("ok")
