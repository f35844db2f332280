"""One K7 shape in a loop (for rocprofv3 --pmc passes): python tools/dev/rs_one.py NIMG A B [acc] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
from uno_amd.resample import _tables
dev = torch.device("cuda:0")
nimg, a, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
acc = len(sys.argv) > 4 and sys.argv[4] == "acc"
n = int(sys.argv[5]) if len(sys.argv) > 5 else 5
x = torch.randn(nimg, a, a, device=dev)
(fh, th), _ = _tables(a, b, str(dev))
(fw, _), _ = _tables(a, b, str(dev))
out = torch.zeros(nimg, b, b, device=dev) if acc else None
for _ in range(n):
    _native.resample2d(x, b, b, fh, fw, th, out=out) if acc else _native.resample2d(x, b, b, fh, fw, th)
torch.cuda.synchronize()
