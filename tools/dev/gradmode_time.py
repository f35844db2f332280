"""Darcy step with the flat gradient buffer (zero fill + autograd's in-place accumulation) against the same step with .grad = None
before every backward (autograd adopts the incoming gradient tensors: no fill, no adds) - the ceiling of what writing the weight
gradients straight into the flat buffer could give.  python tools/dev/gradmode_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch, ComplexAdam, lp_loss_rel_sum
dev = torch.device("cuda:0")
a, u = synthetic_darcy_batch(16, 421, 1234, dev)
def timeit(step, K=20):
    for _ in range(5): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
    return min(ts)
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
print("flat buffer:   %.3f ms/step" % timeit(lambda: tr.step(a, u)))
del tr
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
opt = ComplexAdam(model.parameters(), lr=1e-3, weight_decay=1e-3)
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    loss = lp_loss_rel_sum(model(a).reshape(16, -1), u.reshape(16, -1))
    loss.backward()
    opt.step()
print(".grad = None:  %.3f ms/step" % timeit(step))
