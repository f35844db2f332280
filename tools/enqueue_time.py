# how long does the host need to ENQUEUE one training step vs how long the GPU needs to run it?
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(16, 421, 1234, dev)
for _ in range(3):
    tr.step(a, u)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    tr.step(a, u)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.2f} ms/step   total {1e3*(t2-t0)/K:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    tr.step(a, u)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
