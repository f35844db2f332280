"""K8 / K9 at the channel-mix shapes of one Darcy step (tools/dev/flopcount.py) against max(flops / 157 TF/s, bytes / 5 TB/s):
python tools/dev/cmlayers.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
dev = torch.device("cuda:0")
B = 16
LAYERS = [(3, 32, 177241), (32, 64, 177241), (64, 64, 198916), (64, 128, 49729), (128, 64, 49729), (128, 256, 12321), (256, 256, 12321), (256, 128, 12321)]
tot = floor = 0.0
for Ci, Co, P in LAYERS:
    x = torch.randn(B, Ci, P, device=dev); w = torch.randn(Co, Ci, device=dev); b = torch.randn(Co, device=dev)
    gy = torch.randn(B, Co, P, device=dev)
    fl = max(2.0 * B * Ci * Co * P / 157e12, 4.0 * B * P * (Ci + Co) / 5e12) * 1e6
    t1 = bench._timed(lambda: _native.channel_mix(x, w, b), dev, iters=10, reps=3) * 1e6
    t2 = bench._timed(lambda: _native.channel_mix(gy, w, None, transpose_w=True), dev, iters=10, reps=3) * 1e6
    t3 = bench._timed(lambda: _native.channel_wgrad(gy, x), dev, iters=10, reps=3) * 1e6
    print(f"{Ci:3d}->{Co:3d} P={P:6d}: floor {fl:6.0f} us | fwd {t1:6.0f} ({fl/t1*100:3.0f} %)  dgrad {t2:6.0f} ({fl/t2*100:3.0f} %)  wgrad {t3:6.0f} ({fl/t3*100:3.0f} %)", flush=True)
