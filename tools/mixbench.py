import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import torch
from uno_amd import _native
dev = torch.device("cuda:0")
def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (B, C, S, m) in ((16, 64, 421, 20), (4, 64, 1024, 32), (16, 64, 446, 18)):
    x = torch.randn(B, C, S, S, device=dev); xb = x.bfloat16()
    O = _native.dft2d_forward(x, m, m)
    print(f"S={S} m={m} B={B}: K1 f32 {timed(lambda: _native.dft2d_forward(x, m, m)):7.1f} us  bf16 {timed(lambda: _native.dft2d_forward(xb, m, m)):7.1f} us |"
          f" K3 f32 {timed(lambda: _native.dft2d_inverse(O, S, S)):7.1f} us  bf16 {timed(lambda: _native.dft2d_inverse(O, S, S, dtype=torch.bfloat16)):7.1f} us")
