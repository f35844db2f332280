"""Per-phase cycle stamps of the volume transforms K1v / K3v at the C4 block (development variant:
    python tools/dev/mkvariant.py voldev dft3d_volume.hip -DUNO_VOL_DEV
    python tools/dev/vol3dstamps.py uno_amd/lib/variants/libuno_voldev.so fwd|inv)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
buf = torch.zeros(4096 * 16 * 8, dtype=torch.int64, device=dev)
os.environ["UNO_VOL_STAMPS"] = str(buf.data_ptr())
os.environ["UNO_VOL_WHICH"] = sys.argv[2]
from uno_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
B, C, dims, modes = 8, 32, (64, 64, 20), (16, 16, 8)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, *dims, generator=g).to(dev)
ws = [(0.1 * torch.randn(C, C, *modes, dtype=torch.cfloat, generator=g)).to(dev) for _ in range(4)]
names = ["tables", "wait barrier 1", "phase 1", "wait barrier 2", "phase 2 (issue)", "drain stores"]
for rep in range(4):
    buf.zero_()
    y, xt = _native.spectral_conv3d_forward(x, ws, *dims)
    torch.cuda.synchronize()
v = buf.view(-1, 16, 8)[:B * C].cpu().double()
t0 = v[:, :, 0].min()
print(f"{sys.argv[2]}: first start -> last end {v[:, :, 6].max() - t0:.0f} ticks (100 MHz clock: x10 ns); start spread {v[:, :, 0].max() - t0:.0f}")
for i, nm in enumerate(names):
    d = v[:, :, i + 1] - v[:, :, i]
    print(f"   {nm:22s} mean {d.mean():8.0f}  min {d.min():8.0f}  max {d.max():8.0f}")
d = v[:, :, 6] - v[:, :, 0]
print(f"   {'whole wave':22s} mean {d.mean():8.0f}  min {d.min():8.0f}  max {d.max():8.0f}")
wg = v[:, :, 6].max(1).values - v[:, :, 0].min(1).values
print(f"   {'whole workgroup':22s} mean {wg.mean():8.0f}  min {wg.min():8.0f}  max {wg.max():8.0f}")
end = v[:, :, 6].max(1).values - t0
print("   workgroup end times (ticks after first start), deciles:", [int(q) for q in torch.quantile(end, torch.linspace(0, 1, 11, dtype=torch.double))])
ph = 2 if sys.argv[2] == "fwd" else 4
d = v[:, :, ph + 1] - v[:, :, ph]
print(f"   plane-phase cycles by wave index (mean over workgroups): {[int(t) for t in d.mean(0)]}")
print(f"   plane-phase END (cycles after the workgroup's first start) by wave index: {[int(t) for t in (v[:, :, ph + 1] - v[:, :, 0].min(1, keepdim=True).values).mean(0)]}")
