"""training-step time of the headline workload under a given library: python tools/exp/steptime.py <lib.so|-> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(16, 421, 1234, dev)
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for _ in range(5): tr.step(a, u)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K): loss = tr.step(a, u)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / K * 1e3)
print(f"{sys.argv[1] if len(sys.argv) > 1 else '-':28s} {min(ts):7.3f} ms/step (reps {' '.join('%.3f' % t for t in ts)})  loss {float(loss):.5f}", flush=True)
