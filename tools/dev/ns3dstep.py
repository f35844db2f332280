"""NS-3D training step + per-kernel table under a given library: python tools/dev/ns3dstep.py <lib.so|-> [width]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
from uno_amd.harness import Uno3D_T20, ComplexAdam, ns3d_loss
import uno_amd.integral_operators as _io
if os.environ.get("ONE_BUFFER_3D") == "0":
    _io.ONE_BUFFER_3D = False
w = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
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
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): loss = step()
torch.cuda.synchronize()
print(f"w={w}: {(time.perf_counter()-t0)/5*1e3:.2f} ms/step loss {float(loss):.4f}")
_native.profile_begin(20000)
step()
torch.cuda.synchronize()
agg = {}
for name, ms, by in _native.profile_end():
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"   {k:60s} x{n:4d} {ms*1e3:9.1f} us")
