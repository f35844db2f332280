"""K2 (per-mode complex GEMM) timing for the layer shapes of the three models: python tools/k2bench.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
SHAPES = [  # B, Ci, Co, modes per corner, corners, label
    (16, 64, 128, 18 * 18, 2, "darcy conv1"), (16, 128, 256, 64, 2, "darcy conv2"), (16, 256, 256, 64, 2, "darcy conv3"),
    (16, 256, 64, 18 * 18, 2, "darcy conv5"), (16, 64, 64, 400, 2, "C2 block"),
    (32, 32, 48, 22 * 22, 2, "ns2d L1"), (32, 48, 96, 14 * 14, 2, "ns2d L2"), (32, 96, 192, 36, 2, "ns2d L3"),
    (32, 192, 192, 36, 2, "ns2d L4"), (32, 192, 48, 14 * 14, 2, "ns2d L6"),
    (8, 32, 32, 16 * 16 * 8, 4, "C4 block"),
    (8, 32, 64, 22 * 22 * 5, 4, "ns3d32 L1"), (8, 64, 128, 14 * 14 * 5, 4, "ns3d32 L2"), (8, 128, 256, 180, 4, "ns3d32 L3"),
    (8, 256, 512, 216, 4, "ns3d32 L4"), (8, 512, 128, 216, 4, "ns3d32 L5"),
]
COLD_BYTES = 600e6          # rotate over enough copies of the operands that none of them is still in the 256 MB Infinity Cache
def timeit(fns, reps=30):
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fns[r % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B, Ci, Co, Mc, nc, label in SHAPES:
    wb = 8.0 * Ci * Co * Mc * nc
    ab = 8.0 * B * (Ci + Co) * Mc * nc
    ncopy = max(1, min(12, int(COLD_BYTES / (wb + ab)) + 1))
    xs = [torch.randn(B, Ci, nc, Mc, dtype=torch.cfloat, device=dev) for _ in range(ncopy)]
    gos = [torch.randn(B, Co, nc, Mc, dtype=torch.cfloat, device=dev) for _ in range(ncopy)]
    wss = [[torch.randn(Ci, Co, Mc, dtype=torch.cfloat, device=dev) for _ in range(nc)] for _ in range(ncopy)]
    t0 = timeit([(lambda i=i: _native.mode_mix(xs[i], wss[i], 0)) for i in range(ncopy)])
    t1 = timeit([(lambda i=i: _native.mode_mix(gos[i], wss[i], 1)) for i in range(ncopy)])
    t2 = timeit([(lambda i=i: _native.mode_wgrad(xs[i], gos[i], (Ci, Co, Mc), nc)) for i in range(ncopy)])
    print(f"{label:12s} B={B:2d} {Ci:3d}->{Co:3d} modes {nc}x{Mc:5d}: fwd {t0:6.1f} dgrad {t1:6.1f} wgrad {t2:6.1f} us   "
          f"(weights {wb/1e6:5.1f} MB, act {ab/1e6:5.1f} MB, {ncopy} copies -> {(wb+ab)/t0/1e6:5.2f} TB/s fwd)")
