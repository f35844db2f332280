"""C5-size transforms of bf16 images, a few launches each (for rocprofv3 passes)."""
import sys, os
import torch
sys.path.insert(0, '.')
from uno_amd import _native
dev = torch.device('cuda:0')
n, H, W, m1, m2 = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (256, 1024, 1024, 32, 32))]
xs = [torch.randn(n, 1, H, W, device=dev).bfloat16() for _ in range(2)]
O = torch.randn(n, 1, 2 * m1, m2, dtype=torch.cfloat, device=dev)
for i in range(6):
    _native.dft2d_forward(xs[i % 2], m1, m2)
    _native.dft2d_inverse(O, H, W, dtype=torch.bfloat16)
torch.cuda.synchronize()
