import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for fl in (0, 32):
        env = dict(os.environ, UNO_CM_FLAGS=str(fl))
        subprocess.run([sys.executable, __file__, str(fl)], env=env)
    sys.exit(0)
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
res = []
for (Ci, Co, S) in [(32, 64, 431), (64, 128, 215), (64, 128, 431), (128, 64, 431)]:
    P = S * S
    x = torch.randn(B, Ci, P, device=dev); w = torch.randn(Co, Ci, device=dev); b = torch.randn(Co, device=dev)
    t1 = timeit(lambda: _native.channel_mix(x, w, b))
    res.append(f"{Ci}->{Co}@{S}: {t1*1e3:7.1f} us")
print("flags", sys.argv[1], " | ".join(res), flush=True)
