"""rocprofv3 / PMC target: a few weight-gradient calls of one 1x1 layer.  python tools/dev/wgrad_one.py Ci Co S [B] [mix]  (mix: the forward call instead)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
Ci, Co, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
mix = len(sys.argv) > 5 and sys.argv[5] == "mix"
dev = torch.device("cuda:0")
P = S * S
x = torch.randn(B, Ci, P, device=dev); gy = torch.randn(B, Co, P, device=dev); w = torch.randn(Co, Ci, device=dev)
for _ in range(6):
    if mix: _native.channel_mix(x, w, None)
    else: _native.channel_wgrad(gy, x)
torch.cuda.synchronize()
