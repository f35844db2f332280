"""Darcy step with fc1 - GELU - fc2 on the domain window (crop=) against the same build computing the whole padded grid:
python tools/dev/croptime.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
from uno_amd.harness import models
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
orig = models.channel_mix_cat_project


def run(tag, fn):
    models.channel_mix_cat_project = fn
    torch.manual_seed(0)
    model = UNO_9(3, 64, pad=5).to(dev)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(16, 421, 1234, dev)
    for _ in range(5): tr.step(a, u)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K): loss = tr.step(a, u)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
    print(f"{tag:12s} {min(ts):7.3f} ms/step (reps {' '.join('%.3f' % t for t in ts)})  loss {float(loss):.6f}", flush=True)


for _ in range(2):
    run("whole grid", lambda *a, crop=None, **k: orig(*a, **k))
    run("window", orig)

# per-launch times of both forms on this box (library event pairs)
from uno_amd import _native


def launches(fn):
    models.channel_mix_cat_project = fn
    torch.manual_seed(0)
    model = UNO_9(3, 64, pad=5).to(dev)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(16, 421, 1234, dev)
    for _ in range(5): tr.step(a, u)
    torch.cuda.synchronize()
    runs = []
    for _ in range(5):
        _native.profile_begin(10000)
        tr.step(a, u)
        torch.cuda.synchronize()
        runs.append(_native.profile_end())
    return [(runs[0][i][0], sum(r[i][1] for r in runs) / 5 * 1e3, runs[0][i][2]) for i in range(len(runs[0]))]


la, lb = launches(lambda *a, crop=None, **k: orig(*a, **k)), launches(orig)
print("launches", len(la), len(lb), "sum", sum(v[1] for v in la), sum(v[1] for v in lb))
if len(la) == len(lb):
    for i, (p, q) in enumerate(zip(la, lb)):
        if abs(p[1] - q[1]) > 4 or p[2] != q[2]:
            print(f"{i:3d} {p[0]:44s} {p[1]:7.1f} -> {q[1]:7.1f} us   {p[2] / 1e6:8.1f} -> {q[2] / 1e6:8.1f} MB")
