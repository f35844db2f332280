# one weight-gradient shape, a few launches (for rocprofv3 --pmc): python tools/cw_one.py Ci Co S
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
Ci, Co, S = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(16, Ci, S * S, device=dev); gy = torch.randn(16, Co, S * S, device=dev)
for _ in range(3):
    gw, gb = _native.channel_wgrad(gy, x)
torch.cuda.synchronize()
