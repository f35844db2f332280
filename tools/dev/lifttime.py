"""Darcy step with the three forms of the lift in one process - layer by layer (fc_n1, fc0, GELU + pad pass; everything stored), fc0 with
the padded activation in its epilogue and its result recomputed backward, the whole lift with neither intermediate stored:
python tools/dev/lifttime.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
from uno_amd.harness import models
from uno_amd.integral_operators import channel_mix, gelu_channel_mix, gelu_channel_mix_pad, gelu_pad2d
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fused = models.lift_gelu_pad
two_pass = lambda x, f1, f0, ph, pw: gelu_pad2d(gelu_channel_mix(channel_mix(x, f1.weight, f1.bias), f0.weight, f0.bias), ph, pw)
fc0_only = lambda x, f1, f0, ph, pw: gelu_channel_mix_pad(channel_mix(x, f1.weight, f1.bias), f0.weight, f0.bias, ph, pw)


def setup(fn):
    models.lift_gelu_pad = fn
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
    print(f"{tag:16s} {min(ts):7.3f} ms/step (reps {' '.join('%.3f' % t for t in ts)})  loss {float(loss):.6f}", flush=True)


def launches(fn):
    tr, a, u = setup(fn)
    runs = []
    for _ in range(5):
        _native.profile_begin(10000)
        tr.step(a, u)
        torch.cuda.synchronize()
        runs.append(_native.profile_end())
    return [(runs[0][i][0], sum(r[i][1] for r in runs) / 5 * 1e3, runs[0][i][2]) for i in range(len(runs[0]))]


FORMS = (("layer by layer", two_pass), ("fc0 recomputed", fc0_only), ("whole lift", fused))
for _ in range(2):
    for tag, fn in FORMS:
        run(tag, fn)
for tag, fn in FORMS:
    l = launches(fn)
    print(tag, "sum", sum(v[1] for v in l), " first launches:", " ".join(f"{v[0].replace('uno::', '').replace('_kernel', '')} {v[1]:.0f}" for v in l[:4]), "| last:", " ".join(f"{v[0].replace('uno::', '').replace('_kernel', '')} {v[1]:.0f}" for v in l[-8:-2]))
