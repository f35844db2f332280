"""Which kernel families gain from the alternating sweep direction: step time per mask (uno_sweep_alternation), one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(16, 421, 1234, dev)
for _ in range(5): tr.step(a, u)
masks = [0, 4, 5, 7, 15, 31, 63, 127, 255, 255 - 8, 255 - 16, 255 - 2, 255 - 1, 255 - 128, 4 + 8, 4 + 2, 0, 255]
for rnd in range(2):
    for m in masks:
        _native.lib().uno_sweep_alternation(m if m != 1 else 255)
        for _ in range(2): tr.step(a, u)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): tr.step(a, u)
        torch.cuda.synchronize()
        print(f"round {rnd} mask {m:3d}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
