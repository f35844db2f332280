"""C4 3-D block forward / backward + per-kernel table under a given library: python tools/dev/block3dtime.py <lib.so|->"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
dev = torch.device("cuda:0")
b = bench.spectral_block3d_roofline(dev)
print(sys.argv[1] if len(sys.argv) > 1 else "-", f"fwd {b['fwd_us']:.1f} us ({b['fwd_frac_of_8TBs']*100:.1f} %)  bwd {b['bwd_us']:.1f} us ({b['bwd_frac_of_8TBs']*100:.1f} %)")
for k, v in b["fwd_kernels"].items(): print("   fwd", k, f"{v['avg_us']:.1f}")
for k, v in b["bwd_kernels"].items(): print("   bwd", k, f"{v['avg_us']:.1f}")
