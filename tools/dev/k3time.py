"""K3 (and K1) at the C2 size under an experiment library: python tools/exp/k3time.py <lib.so> [S] [m]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
S = int(sys.argv[2]) if len(sys.argv) > 2 else 421
m = int(sys.argv[3]) if len(sys.argv) > 3 else 20
B, C = 16, 64
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, S, S, generator=g).to(dev)
X = _native.dft2d_forward(x, m, m, scale=1.0 / (S * S))
def timeit(fn, n=20, warm=3, reps=5):
    for _ in range(warm): fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n)
    out.sort(); return out[len(out) // 2]
t1 = timeit(lambda: _native.dft2d_forward(x, m, m, scale=1.0 / (S * S))) * 1e3
t3 = timeit(lambda: _native.dft2d_inverse(X, S, S)) * 1e3
gb = B * C * S * S * 4 / 1e9
tag = os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "product"
env = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("EXP_"))
print(f"{tag:16s} {env:24s} S={S} K1 {t1:7.1f} us ({gb/t1*1e3:5.2f} TB/s)   K3 {t3:7.1f} us ({gb/t3*1e3:5.2f} TB/s)", flush=True)
