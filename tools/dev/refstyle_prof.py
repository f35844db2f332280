"""rocprofv3 target: a few training steps of the reference-style caller (harness/reference_style.py) on the product blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd.harness import DarcyTrainer, UNO_9_ReferenceStyle, synthetic_darcy_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNO_9_ReferenceStyle(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(16, 421, 1234, dev)
for _ in range(6):
    tr.step(a, u)
torch.cuda.synchronize()
