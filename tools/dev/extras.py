"""bench.py's secondary workloads alone under a given library: python tools/dev/extras.py <lib.so|->"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
dev = torch.device("cuda:0")
if "--block3d" in sys.argv:
    b = bench.spectral_block3d_roofline(dev)
    print("block3d", round(b["fwd_us"], 1), round(b["bwd_us"], 1))
for k, v in bench.extra_workloads(dev).items():
    print(k, v.get("ms_per_step"), v.get("error"))
