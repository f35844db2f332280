import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
dev = torch.device("cuda:0")
def timeit(fn, n=10, reps=3):
    for _ in range(2): fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1) / n)
    return sorted(out)[len(out)//2]
for (B, C, S) in [(16, 256, 111), (16, 128, 223)]:
    x = torch.randn(B, C, S, S, device=dev); g = torch.randn(C, device=dev); b = torch.randn(C, device=dev); gy = torch.randn_like(x)
    y, m, r = _native.instnorm_forward(x, g, b, 1e-5, True)
    t1 = timeit(lambda: _native.instnorm_forward(x, g, b, 1e-5, True))
    t2 = timeit(lambda: _native.instnorm_backward(x, gy, g, b, m, r, True))
    by = x.numel() * 4
    print(f"{B}x{C}x{S}^2: fwd {t1*1e3:7.1f} us ({2*by/t1/1e9:5.2f} TB/s)  bwd {t2*1e3:7.1f} us ({3*by/t2/1e9:5.2f} TB/s)", flush=True)
