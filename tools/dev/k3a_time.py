"""K3-A kernel timing only: python tools/dev/k3a_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native, resample as rs
dev = torch.device("cuda:0")
def timed(fn, iters=20, reps=5, warm=3):
    for _ in range(warm): fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    out.sort(); return out[len(out) // 2]
g = torch.Generator().manual_seed(0)
for (B, C, Hs, H, m) in ((16, 64, 223, 446, 18), (16, 128, 111, 223, 8)):
    spec = torch.randn(B, C, 2 * m, m, dtype=torch.complex64, generator=g).to(dev)
    t = torch.randn(B, C, Hs, Hs, generator=g).to(dev)
    tabs = rs.upsample_add_tables(Hs, Hs, H, H, str(dev), False)
    k3 = timed(lambda: _native.dft2d_inverse(spec, H, H, 1.0, True, True))
    k3a = timed(lambda: _native.dft2d_inverse(spec, H, H, 1.0, True, True, addend=(t, tabs)))
    print(f"stagger {os.environ.get('UNO_K3A_STAGGER','0')}: {H}^2 K3 {k3:.1f} us K3-A {k3a:.1f} us", flush=True)
