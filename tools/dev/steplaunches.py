"""Every library launch of ONE Darcy training step in issue order: kernel, us, algorithmic MB, GB/s (library event pairs; averaged
over a few steps).  python tools/dev/steplaunches.py [lib.so|-]"""
import os, sys
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
for _ in range(5): tr.step(a, u)
torch.cuda.synchronize()
N = 5
runs = []
for _ in range(N):
    _native.profile_begin(10000)
    tr.step(a, u)
    torch.cuda.synchronize()
    runs.append(_native.profile_end())
n = len(runs[0])
assert all(len(r) == n for r in runs)
tot = 0.0
for i in range(n):
    name, by = runs[0][i][0], runs[0][i][2]
    us = sum(r[i][1] for r in runs) / N * 1e3
    tot += us
    print(f"{i:3d} {name.replace('uno::',''):48s} {us:8.1f} us {by/1e6:9.1f} MB {by/us/1e3:7.0f} GB/s")
print("sum", tot, "us")
