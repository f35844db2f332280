"""Per-phase cycle stamps of K8-S (CT = 128, f32; NO_SHADOW=1: weights split per workgroup) for one layer shape, from a -DUNO_CMS_DEV variant:
python tools/dev/cmsstamps.py <variant.so> Ci Co P [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
lib, Ci, Co, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dev = torch.device("cuda:0")
npt, ncot = (P + 127) // 128, Co // 128
grid = 8 * ((npt * ncot + 7) // 8)
buf = torch.zeros(B * grid * 4 * 8, dtype=torch.int64, device=dev)
os.environ["UNO_CMS_EXP"] = "64"
os.environ["UNO_CMS_STAMPS"] = hex(buf.data_ptr())
from uno_amd import _native
_native.LIB_PATH = os.path.abspath(lib)
if os.environ.get("NO_SHADOW"):          # weights split per workgroup (WM = 0) instead of the pre-split form
    _native._mix_scratch.__init__ = lambda self, device, *a: (setattr(self, "bytes", 0), setattr(self, "device", device), setattr(self, "buf", None))[0]
x = torch.randn(B, Ci, P, device=dev)
w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5)
bias = torch.randn(Co, device=dev)
for _ in range(3):
    buf.zero_()
    y = _native.channel_mix(x, w, bias)
    torch.cuda.synchronize()
s = buf.view(B * grid, 4, 8).double()
live = s[:, 0, 5] > 0
s = s[live]
names = ["store (waits for loads + split + LDS writes)", "barrier 1", "compute (LDS reads + MFMA)", "barrier 2", "epilogue", "whole K loop"]
print(f"{Ci} -> {Co}, P = {P}, B = {B}: {int(live.sum())} workgroups, {Ci // 32} chunks")
for i, n in enumerate(names):
    print(f"  {n:46s} {s[:, :, i].mean().item():9.0f} cycles per wave (min {s[:, :, i].min().item():.0f}, max {s[:, :, i].max().item():.0f})")
t0 = s[:, 0, 6]
life = s[:, :, 5].mean(1) + s[:, :, 4].mean(1)
print(f"  K loop + epilogue per workgroup {life.mean().item():9.0f} cycles; kernel span {(s[:, :, 7].max() - s[:, :, 7].min()).item() / 100:.1f} us (100 MHz clock)")
