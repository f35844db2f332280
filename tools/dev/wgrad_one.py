"""rocprofv3 / PMC target: a few weight-gradient calls of one 1x1 layer.  python tools/dev/wgrad_one.py Ci Co S [B] [mix] [lib.so]
(mix: the forward call instead; lib.so: a library variant, e.g. the cycle-stamped build `tools/dev/mkvariant.py cwsstamps channel_mix.hip -DUNO_CWS_STAMPS`)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
for _a in sys.argv[1:]:
    if _a.endswith('.so'): _native.LIB_PATH = os.path.abspath(_a)
Ci, Co, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
mix = "mix" in sys.argv[5:]
dev = torch.device("cuda:0")
P = S * S
x = torch.randn(B, Ci, P, device=dev); gy = torch.randn(B, Co, P, device=dev); w = torch.randn(Co, Ci, device=dev)
for _ in range(6):
    if mix: _native.channel_mix(x, w, None)
    else: _native.channel_wgrad(gy, x)
torch.cuda.synchronize()
