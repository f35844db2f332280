"""K1p / K3p (plane-batched pruned DFTs) timing vs image count: python tools/planebench.py [H W m1 m2 [image counts ...]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
dev = torch.device("cuda:0")
H, W, m1, m2 = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 20, 16, 8)
NS = tuple(int(a) for a in sys.argv[5:]) or (256, 1024, 4096, 16384, 65536)
for n in NS:
    x = torch.randn(n, 1, H, W, device=dev)
    O = _native.dft2d_forward(x, m1, m2)
    for name, fn, by in (("fwd", lambda: _native.dft2d_forward(x, m1, m2), 0), ("inv", lambda: _native.dft2d_inverse(O, H, W), 0)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        gb = n * (H * W * 4 + 2 * m1 * m2 * 8) / 1e9
        print(f"n={n:6d} {name}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  ({us / n * 1e3:6.1f} ns/image)")
