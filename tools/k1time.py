import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
B, C, S, m = 16, 64, int(os.environ.get('S', 421)), 20
dev = torch.device("cuda:0")
x = torch.randn(B, C, S, S, device=dev)
O = torch.randn(B, C, 2 * m, m, dtype=torch.cfloat, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t1 = timeit(lambda: _native.dft2d_forward(x, m, m))
t3 = timeit(lambda: _native.dft2d_inverse(O, S, S))
gb = B * C * S * S * 4 / 1e9
print(f"{'product':50s} S={S} K1 {t1:7.1f} us ({gb/t1*1e3:5.2f} TB/s)   K3 {t3:7.1f} us ({gb/t3*1e3:5.2f} TB/s)")
