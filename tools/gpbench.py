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
B, C, S = 16, 64, 446
pre = torch.randn(B, C, S * S, device=dev); w = torch.randn(C, device=dev); b = torch.randn(1, device=dev); go = torch.randn(B, S * S, device=dev)
t1 = timeit(lambda: _native.gelu_project_forward(pre, w, b))
t2 = timeit(lambda: _native.gelu_project_backward(pre, w, go))
by = pre.numel() * 4
print(f"gelu_project {B}x{C}x{S}^2: fwd {t1*1e3:7.1f} us ({by/t1/1e9:5.2f} TB/s)  bwd {t2*1e3:7.1f} us ({2*by/t2/1e9:5.2f} TB/s)")
s = torch.randn(B, C, 421, 421, device=dev); gy = torch.randn(B, C, 446, 446, device=dev)
t3 = timeit(lambda: _native.gelu_pad(s, 446, 446)); t4 = timeit(lambda: _native.gelu_pad_backward(s, gy))
print(f"gelu_pad fwd {t3*1e3:7.1f} us ({(s.numel()+gy.numel())*4/t3/1e9:5.2f} TB/s)  bwd {t4*1e3:7.1f} us ({3*s.numel()*4/t4/1e9:5.2f} TB/s)")
