"""Every library launch of ONE training step of a secondary workload in issue order (kernel, us, algorithmic MB, GB/s):
python tools/dev/launches_of.py c5|c5mixed|ns3d8|ns3d32|ns2d1   (ns2d1: one roll-out step forward + backward, eager)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
from uno_amd.harness import (ComplexAdam, DarcyTrainer, MixedDarcyTrainer, UNO, UNO_9, Uno3D_T20, ns2d_rollout_loss, ns3d_loss,
                             synthetic_darcy_batch)
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
dev = torch.device("cuda:0")
torch.manual_seed(0)
if which in ("c5", "c5mixed"):
    model = UNO_9(3, 64, pad=5).to(dev)
    tr = (MixedDarcyTrainer if which == "c5mixed" else DarcyTrainer)(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(4, 1024, 1234, dev)
    step = lambda: tr.step(a, u)
elif which.startswith("ns3d"):
    m3 = Uno3D_T20(6, int(which[4:]), pad=3).to(dev)
    x, y = torch.randn(8, 64, 64, 10, 1, device=dev), torch.randn(8, 64, 64, 20, device=dev)
    opt = ComplexAdam(m3.parameters(), lr=1e-3, weight_decay=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = ns3d_loss(m3, x, y)
        loss.backward()
        opt.step()
else:
    m = UNO(14, 32).to(dev)
    xx, yy = torch.randn(32, 64, 64, 10, device=dev), torch.randn(32, 64, 64, 40, device=dev)
    opt = ComplexAdam(m.parameters(), lr=1e-3, weight_decay=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        ns2d_rollout_loss(m, xx, yy, T_f=1, step=1).backward()
        opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 3
runs = []
for _ in range(N):
    _native.profile_begin(20000)
    step()
    torch.cuda.synchronize()
    runs.append(_native.profile_end())
n = len(runs[0])
tot = 0.0
for i in range(n):
    name, by = runs[0][i][0], runs[0][i][2]
    us = sum(r[i][1] for r in runs if len(r) == n) / sum(1 for r in runs if len(r) == n) * 1e3
    tot += us
    print(f"{i:3d} {name.replace('uno::',''):52s} {us:8.1f} us {by/1e6:9.1f} MB {by/us/1e3:7.0f} GB/s")
print("sum", tot, "us")
