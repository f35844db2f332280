"""Per-phase cycle stamps of K16 (lift_forward_kernel) at the Darcy shape from a -DUNO_LB_DEV variant: python tools/dev/lfstamps.py <variant.so>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
B, H, W, pad = 16, 421, 421, 25
npt = (H * ((W + pad) // 4 * 4) + 127) // 128
nwg = (npt + 7) // 8
buf = torch.zeros(B * nwg * 4 * 8, dtype=torch.int64, device=dev)
os.environ["UNO_LF_STAMPS"] = hex(buf.data_ptr())
from uno_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
torch.manual_seed(0)
x = torch.randn(B, 3, H, W, device=dev)
w1, b1 = torch.randn(32, 3, device=dev), torch.randn(32, device=dev)
w0, b0 = torch.randn(64, 32, device=dev) / 6, torch.randn(64, device=dev)
for _ in range(3):
    buf.zero_()
    _native.lift_forward(x, w1, b1, w0, b0, H + pad, W + pad)
    torch.cuda.synchronize()
s = buf.view(-1, 4, 8).double()
names = ["phase 0: a -> LDS, next x issued", "barriers", "first half's MFMAs", "second half's MFMAs + both GELU epilogues", "-", "-", "-", "stores"]
tot = s.sum(2).mean().item()
for i, n in enumerate(names):
    if n != "-":
        print(f"  {n:48s} {s[:, :, i].mean().item() / 8:9.0f} cycles per wave and tile ({100 * s[:, :, i].mean().item() / tot:4.1f} %)")
print(f"  per workgroup (8 tiles) {tot:9.0f} cycles")
