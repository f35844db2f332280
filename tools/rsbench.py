import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
from uno_amd.resample import _tables
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
for (nimg, a, b) in [(1024, 446, 334), (2048, 334, 223), (1024, 446, 223), (2048, 223, 111), (2048, 111, 223), (1024, 223, 446)]:
    x = torch.randn(nimg, a, a, device=dev)
    (fh, th), (bh, tbh) = _tables(a, b, str(dev))
    (fw, _), (bw, _) = _tables(a, b, str(dev))
    y = _native.resample2d(x, b, b, fh, fw, th)
    t = timeit(lambda: _native.resample2d(x, b, b, fh, fw, th))
    by = 4 * nimg * (a * a + b * b)
    tb = timeit(lambda: _native.resample2d(y, a, a, bh, bw, tbh))
    t2 = timeit(lambda: _native.resample2d(x, b, b, fh, fw, None))
    print(f"{nimg} x {a}->{b}: fwd {t*1e3:7.1f} us {by/t/1e9:5.2f} TB/s (NP={th[1].shape[1]}) | adjoint {tb*1e3:7.1f} us {by/tb/1e9:5.2f} TB/s (NP={tbh[1].shape[1]}) | two-pass fwd {t2*1e3:7.1f} us")
print("accumulating calls (out += ...):")
for (nimg, a, b) in [(1024, 223, 446), (2048, 111, 223), (1024, 446, 223), (1024, 334, 446), (2048, 223, 334)]:
    x = torch.randn(nimg, a, a, device=dev)
    (fh, th), _ = _tables(a, b, str(dev))
    (fw, _), _ = _tables(a, b, str(dev))
    out = torch.zeros(nimg, b, b, device=dev)
    t = timeit(lambda: _native.resample2d(x, b, b, fh, fw, th, out=out))
    by = 4 * nimg * (a * a + 2 * b * b)
    print(f"{nimg} x {a}->{b} accumulate: {t*1e3:7.1f} us {by/t/1e9:5.2f} TB/s")
