"""K2 experiments (development variant: python tools/dev/mkvariant.py k2dev mode_gemm.hip -DUNO_K2_DEV): forward / input-gradient /
weight-gradient time of a few layer shapes with cold operands, under UNO_K2_REMAP / UNO_K2_SHADOW / UNO_K2_KS.
    python tools/dev/k2dev.py <lib.so> [shape label substring]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
SHAPES = [  # B, Ci, Co, modes per corner, corners, label
    (16, 64, 128, 18 * 18, 2, "darcy conv1"), (16, 128, 256, 64, 2, "darcy conv2"), (16, 256, 256, 64, 2, "darcy conv3"),
    (16, 256, 128, 64, 2, "darcy conv4"), (16, 256, 64, 18 * 18, 2, "darcy conv5"), (16, 64, 64, 400, 2, "C2 block"),
    (4, 64, 64, 1024, 2, "C5 block"),
    (32, 32, 48, 22 * 22, 2, "ns2d L1"), (32, 48, 96, 14 * 14, 2, "ns2d L2"), (32, 96, 192, 36, 2, "ns2d L3"),
    (32, 192, 192, 36, 2, "ns2d L4"), (32, 192, 48, 14 * 14, 2, "ns2d L6"),
    (8, 32, 32, 16 * 16 * 8, 4, "C4 block"),
    (8, 32, 64, 22 * 22 * 5, 4, "ns3d32 L1"), (8, 64, 128, 14 * 14 * 5, 4, "ns3d32 L2"), (8, 128, 256, 180, 4, "ns3d32 L3"),
    (8, 256, 512, 216, 4, "ns3d32 L4"), (8, 512, 128, 216, 4, "ns3d32 L5"),
    (8, 8, 16, 22 * 22 * 5, 4, "ns3d8 L1"), (8, 64, 128, 216, 4, "ns3d8 L4"),
]
def timeit(fns, reps=40):
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
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("UNO_K2"))
for B, Ci, Co, Mc, nc, label in SHAPES:
    if want not in label:
        continue
    wb = 8.0 * Ci * Co * Mc * nc
    ab = 8.0 * B * (Ci + Co) * Mc * nc
    ncopy = max(1, min(12, int(600e6 / (wb + ab)) + 1))
    xs = [torch.randn(B, Ci, nc, Mc, dtype=torch.cfloat, device=dev) for _ in range(ncopy)]
    gos = [torch.randn(B, Co, nc, Mc, dtype=torch.cfloat, device=dev) for _ in range(ncopy)]
    wss = [[torch.randn(Ci, Co, Mc, dtype=torch.cfloat, device=dev) for _ in range(nc)] for _ in range(ncopy)]
    def relerr(a, b):
        return float((a.to(torch.complex128) - b).abs().max() / b.abs().max())
    W = torch.stack(wss[0], 2).to(torch.complex128)            # (Ci, Co, nc, Mc)
    e0 = relerr(_native.mode_mix(xs[0], wss[0], 0), torch.einsum("bicq,iocq->bocq", xs[0].to(torch.complex128), W))
    e1 = relerr(_native.mode_mix(gos[0], wss[0], 1), torch.einsum("bocq,iocq->bicq", gos[0].to(torch.complex128), W.conj()))
    gws = _native.mode_wgrad(xs[0], gos[0], (Ci, Co, Mc), nc)
    e2 = relerr(torch.stack(list(gws), 2), torch.einsum("bicq,bocq->iocq", xs[0].to(torch.complex128).conj(), gos[0].to(torch.complex128)))
    bad = "" if max(e0, e1, e2) < 1e-5 else f"   !!! WRONG RESULTS: errors {e0:.1e} {e1:.1e} {e2:.1e}"
    t0 = timeit([(lambda i=i: _native.mode_mix(xs[i], wss[i], 0)) for i in range(ncopy)])
    t1 = timeit([(lambda i=i: _native.mode_mix(gos[i], wss[i], 1)) for i in range(ncopy)])
    t2 = 0.0
    if True:
        t2 = timeit([(lambda i=i: _native.mode_wgrad(xs[i], gos[i], (Ci, Co, Mc), nc)) for i in range(ncopy)])
    print(f"[{tag}] {label:12s}: fwd {t0:6.1f} ({(wb+ab)/t0/1e6:4.2f} TB/s)  dgrad {t1:6.1f} ({(wb+ab)/t1/1e6:4.2f})  wgrad {t2:6.1f} ({(wb+ab)/max(t2,1e-9)/1e6:4.2f}){bad}", flush=True)
