import sys, os
import numpy as np, torch
sys.path.insert(0, '.')
from uno_amd import _native
dev = torch.device('cuda:0')
os.makedirs('gpurun_out/dbg', exist_ok=True)
for W in (257, 258, 259, 289, 513):
    g = torch.Generator().manual_seed(W)
    x = torch.randn(1, 1, 16, W, generator=g).bfloat16()
    got = _native.dft2d_forward(x.to(dev), 4, 4).cpu().numpy()
    np.save(f'gpurun_out/dbg/x_{W}.npy', x.float().numpy())
    np.save(f'gpurun_out/dbg/got_{W}.npy', got)
