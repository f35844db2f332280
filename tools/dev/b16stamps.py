"""Per-phase cycle stamps of the bf16-MFMA transforms (development: UNO_B16_*_EXP bit 16).  usage: b16stamps.py fwd|inv [n H W m1 m2]"""
import os, sys
import torch
which = sys.argv[1]
n, H, W, m1, m2 = [int(v) for v in (sys.argv[2:7] if len(sys.argv) > 6 else (256, 1024, 1024, 32, 32))]
dev = torch.device('cuda:0')
buf = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device=dev)
os.environ["UNO_B16_STAMPS"] = str(buf.data_ptr())
os.environ["UNO_B16_FWD_EXP" if which == "fwd" else "UNO_B16_INV_EXP"] = str(16 + int(os.environ.get("EXTRA", "0")))
sys.path.insert(0, '.')
from uno_amd import _native
if os.environ.get('UNO_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['UNO_LIB'])
x = torch.randn(n, 1, H, W, device=dev).bfloat16()
O = torch.randn(n, 1, 2 * m1, m2, dtype=torch.cfloat, device=dev)
for _ in range(int(os.environ.get('REPS', '3'))):
    if which == "fwd":
        _native.dft2d_forward(x, m1, m2)
    else:
        _native.dft2d_inverse(O, H, W, dtype=torch.bfloat16)
torch.cuda.synchronize()
v = buf.view(-1, 4).cpu().double()
v = v[(v.sum(1) > 0)]
names = ["rotate", "row stage", "column stage", "clock (cycles per 10 us / 1000 = 0.1 MHz)"] if which == "fwd" else ["column stage", "split", "row stage", "flush"]
tot = v.sum(1).mean()
print(f"{which} {n}x{H}x{W} m=({m1},{m2}): waves {len(v)}, cycles per wave {tot:.0f} (= {tot/2.1e3:.1f} us at 2.1 GHz)")
for i, nm in enumerate(names):
    print(f"   {nm:16s} {v[:, i].mean():10.0f} cycles  {100 * v[:, i].mean() / tot:5.1f} %   (min {v[:, i].min():.0f}, max {v[:, i].max():.0f})")
