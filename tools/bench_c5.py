# Config C5 grid size in f32 (the mixed-precision part of C5 is not built): UNO_9(3, 64, pad=5) at 1024^2 (padded 1089^2), and the
# block SpectralConv2d(64, 64, 1024, 1024, 32, 32) - does the path hold at that size, and how fast is it.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(B, 1024, 1234, dev)
for _ in range(2):
    loss = tr.step(a, u)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    loss = tr.step(a, u)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
assert float(loss) == float(loss)
print(f"UNO_9(3,64,pad=5) 1024^2 B={B}: {dt*1e3:.1f} ms/step  {B/dt:.1f} samples/s  loss {float(loss):.4f}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
del tr, model
g = torch.Generator().manual_seed(0)
C, S, m = 64, 1024, 32
x = torch.randn(B, C, S, S, generator=g).to(dev)
w1 = (0.1 * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev); w2 = w1.clone()
gy = torch.randn(B, C, S, S, generator=g).to(dev)
y, xt = _native.spectral_conv2d_forward(x, w1, w2, S, S)
def timed(fn, n=5):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
tf = timed(lambda: _native.spectral_conv2d_forward(x, w1, w2, S, S))
tb = timed(lambda: _native.spectral_conv2d_backward(gy, xt, w1, w2, S, S))
img = B * C * S * S * 4; wb = 2 * C * C * m * m * 8
print(f"SpectralConv2d(64,64,1024,1024,32,32) B={B}: fwd {tf*1e6:.0f} us ({(2*img+wb)/tf/1e12:.2f} TB/s = {(2*img+wb)/tf/8e12*100:.1f} %)  bwd {tb*1e6:.0f} us ({(2*img+2*wb)/tb/8e12*100:.1f} %)")
