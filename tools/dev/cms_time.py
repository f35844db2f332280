"""K8-S at one layer shape: python tools/dev/cms_time.py Ci Co P [B]  (UNO_CMS_EXP / UNO_CM_SPLIT_OFF in the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if os.environ.get('UNO_LIB'): _native.LIB_PATH = os.path.abspath(os.environ['UNO_LIB'])
import bench
dev = torch.device("cuda:0")
Ci, Co, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
x = torch.randn(B, Ci, P, device=dev); w = torch.randn(Co, Ci, device=dev); b = torch.randn(Co, device=dev)
gy = torch.randn(B, Co, P, device=dev)
t1 = bench._timed(lambda: _native.channel_mix(x, w, b), dev, iters=10, reps=3) * 1e6
t2 = bench._timed(lambda: _native.channel_mix(gy, w, None, transpose_w=True), dev, iters=10, reps=3) * 1e6
print(f"exp={os.environ.get('UNO_CMS_EXP', '0'):>3s} off={os.environ.get('UNO_CM_SPLIT_OFF', '-')} {Ci}->{Co} P={P}: fwd {t1:6.1f} us  dgrad {t2:6.1f} us", flush=True)
