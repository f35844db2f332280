"""Darcy step with the lift's GELU + padding written by fc0's epilogue against the same build running the separate GELU + pad pass:
python tools/dev/lifttime.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
from uno_amd.harness import models
from uno_amd.integral_operators import gelu_channel_mix, gelu_pad2d
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fused = models.gelu_channel_mix_pad
two_pass = lambda pre, w, b, ph, pw: gelu_pad2d(gelu_channel_mix(pre, w, b), ph, pw)


def setup(fn):
    models.gelu_channel_mix_pad = fn
    torch.manual_seed(0)
    model = UNO_9(3, 64, pad=5).to(dev)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(16, 421, 1234, dev)
    for _ in range(5): tr.step(a, u)
    torch.cuda.synchronize()
    return tr, a, u


def run(tag, fn):
    tr, a, u = setup(fn)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K): loss = tr.step(a, u)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
    print(f"{tag:12s} {min(ts):7.3f} ms/step (reps {' '.join('%.3f' % t for t in ts)})  loss {float(loss):.6f}", flush=True)


def launches(fn):
    tr, a, u = setup(fn)
    runs = []
    for _ in range(5):
        _native.profile_begin(10000)
        tr.step(a, u)
        torch.cuda.synchronize()
        runs.append(_native.profile_end())
    return [(runs[0][i][0], sum(r[i][1] for r in runs) / 5 * 1e3, runs[0][i][2]) for i in range(len(runs[0]))]


for _ in range(2):
    run("two passes", two_pass)
    run("fused", fused)
for tag, fn in (("two passes", two_pass), ("fused", fused)):
    l = launches(fn)
    print(tag, "sum", sum(v[1] for v in l), " first launches:", " ".join(f"{v[0].replace('uno::', '').replace('_kernel', '')} {v[1]:.0f}" for v in l[:4]))
