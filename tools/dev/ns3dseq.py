"""NS-3D steps the way bench.py's extras run them (w = 8 then w = 32 in one process, 2 warm-up + 5 timed): python tools/dev/ns3dseq.py <lib.so|->"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
from uno_amd.harness import Uno3D_T20, ComplexAdam, ns3d_loss
dev = torch.device("cuda:0")
def run(w, warm):
    torch.manual_seed(0)
    m3 = Uno3D_T20(6, w, pad=3).to(dev)
    x, y = torch.randn(8, 64, 64, 10, 1, device=dev), torch.randn(8, 64, 64, 20, device=dev)
    opt = ComplexAdam(m3.parameters(), lr=1e-3, weight_decay=1e-4)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = ns3d_loss(m3, x, y)
        loss.backward()
        opt.step()
        return loss
    for _ in range(warm): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"w={w} warm={warm}: " + " ".join(f"{t:.1f}" for t in ts), flush=True)
    torch.cuda.empty_cache()
run(8, 2); run(32, 2); run(32, 2); run(8, 2)
