# one channel-mix shape, a few launches (for rocprofv3 --pmc): python tools/cm_one.py Ci Co S
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
Ci, Co, S = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(16, Ci, S * S, device=dev); w = torch.randn(Co, Ci, device=dev); b = torch.randn(Co, device=dev)
for _ in range(3):
    y = _native.channel_mix(x, w, b)
torch.cuda.synchronize()
