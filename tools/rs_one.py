# one resample shape, a few launches (for rocprofv3 --pmc): python tools/rs_one.py nimg a b
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd.resample import resample_forward
nimg, a, b = (int(v) for v in sys.argv[1:4])
x = torch.randn(nimg, a, a, device="cuda")
for _ in range(3):
    y = resample_forward(x, b, b)
torch.cuda.synchronize()
