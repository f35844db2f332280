import os, sys
sys.path.insert(0, os.getcwd())
import torch
from uno_amd import _native
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (B, C, P) in ((32, 128, 4096), (8, 128, 64 * 64 * 26), (32, 64, 4096)):
    pre = torch.randn(B, C, P, device=dev); w = torch.randn(C, device=dev); b = torch.randn(1, device=dev); go = torch.randn(B, P, device=dev)
    tf = timeit(lambda: _native.gelu_project_forward(pre, w, b)); tb = timeit(lambda: _native.gelu_project_backward(pre, w, go))
    print(f"gelu_project B={B} C={C} P={P}: fwd {tf:6.1f} us  bwd {tb:6.1f} us")
