"""rocprofv3 target: a few mixed-precision training steps of the C5 model (UNO_9(3,64,pad=5) at 1024^2, batch 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd.harness import UNO_9, synthetic_darcy_batch
from uno_amd.harness.mixed import MixedDarcyTrainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = MixedDarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(4, 1024, 1234, dev)
for _ in range(6):
    tr.step(a, u)
torch.cuda.synchronize()
