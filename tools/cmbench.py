# channel-mix kernels vs rocBLAS (baddbmm / einsum) on the shapes of the Darcy UNO_9 step
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
B = 16
for (Ci, Co, S) in [(3, 32, 431), (32, 64, 431), (64, 128, 215), (128, 256, 107), (256, 256, 53), (256, 128, 107), (256, 64, 215), (128, 64, 431), (64, 128, 431), (128, 1, 431), (64, 128, 323), (128, 128, 215)]:
    P = S * S
    x = torch.randn(B, Ci, P, device=dev); w = torch.randn(Co, Ci, device=dev); b = torch.randn(Co, device=dev)
    gy = torch.randn(B, Co, P, device=dev)
    by = 4.0 * B * P * (Ci + Co)
    t1 = timeit(lambda: _native.channel_mix(x, w, b))
    t1r = timeit(lambda: torch.baddbmm(b.view(1, -1, 1), w.unsqueeze(0).expand(B, -1, -1), x))
    t2 = timeit(lambda: _native.channel_mix(gy, w, None, transpose_w=True))
    t2r = timeit(lambda: torch.matmul(w.t(), gy))
    t3 = timeit(lambda: _native.channel_wgrad(gy, x))
    t3r = timeit(lambda: (torch.einsum("bop,bip->oi", gy, x), gy.sum(dim=(0, 2))))
    print(f"Ci={Ci:3d} Co={Co:3d} P={S}^2: fwd {t1*1e3:7.1f} us ({by/t1/1e9:5.2f} TB/s) rocblas {t1r*1e3:7.1f} | dgrad {t2*1e3:7.1f} ({by/t2/1e9:5.2f}) rocblas {t2r*1e3:7.1f} | wgrad {t3*1e3:7.1f} ({by/t3/1e9:5.2f}) rocblas {t3r*1e3:7.1f}", flush=True)
