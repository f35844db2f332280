"""Darcy step with the wide 1x1 layers on pre-split weights (scratch provided) against the same build splitting them per workgroup:
python tools/dev/shadowtime.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
orig_init = _native._mix_scratch.__init__


def no_scratch(self, device, Ci, Co, P, bf16):
    self.bytes, self.device, self.buf = 0, device, None


def setup():
    torch.manual_seed(0)
    model = UNO_9(3, 64, pad=5).to(dev)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(16, 421, 1234, dev)
    for _ in range(5): tr.step(a, u)
    torch.cuda.synchronize()
    return tr, a, u


def run(tag, init):
    _native._mix_scratch.__init__ = init
    tr, a, u = setup()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K): loss = tr.step(a, u)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
    print(f"{tag:12s} {min(ts):7.3f} ms/step (reps {' '.join('%.3f' % t for t in ts)})  loss {float(loss):.6f}", flush=True)


def launches(init):
    _native._mix_scratch.__init__ = init
    tr, a, u = setup()
    runs = []
    for _ in range(5):
        _native.profile_begin(10000)
        tr.step(a, u)
        torch.cuda.synchronize()
        runs.append(_native.profile_end())
    return [(runs[0][i][0], sum(r[i][1] for r in runs) / 5 * 1e3, runs[0][i][2]) for i in range(len(runs[0]))]


for _ in range(2):
    run("per tile", no_scratch)
    run("pre-split", orig_init)
la, lb = launches(no_scratch), launches(orig_init)
print("launches", len(la), len(lb), "sum", sum(v[1] for v in la), sum(v[1] for v in lb))
if len(la) == len(lb):
    for i, (p, q) in enumerate(zip(la, lb)):
        if abs(p[1] - q[1]) > 4:
            print(f"{i:3d} {p[0]:44s} {p[1]:7.1f} -> {q[1]:7.1f} us")
