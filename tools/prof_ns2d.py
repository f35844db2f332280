import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd.harness import UNO, ComplexAdam, ns2d_rollout_loss
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = UNO(14, 32).to(dev)
opt = ComplexAdam(m.parameters(), lr=1e-3, weight_decay=1e-4)
xx = torch.randn(32, 64, 64, 10, device=dev); yy = torch.randn(32, 64, 64, 40, device=dev)
for i in range(4):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = ns2d_rollout_loss(m, xx, yy, T_f=40, step=1)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
torch.cuda.synchronize()
print(f"last step: enqueue fwd {1e3*(t1-t0):.1f} ms, bwd {1e3*(t2-t1):.1f} ms, total wall {1e3*(time.perf_counter()-t0):.1f} ms")
