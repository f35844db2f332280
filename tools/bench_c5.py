# Config C5: UNO_9(3, 64, pad=5) at 1024^2 (padded 1089^2) in f32 (the whole-model mixed precision is not built), and the block
# SpectralConv2d(64, 64, 1024, 1024, 32, 32) in f32 and in its mixed-precision form (bf16 activations, f32 accumulation).
# `python tools/bench_c5.py [B] [block]`: 'block' skips the model step.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
BLOCK_ONLY = len(sys.argv) > 2 and sys.argv[2] == "block"
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
if not BLOCK_ONLY:
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
    del tr
del model
g = torch.Generator().manual_seed(0)
C, S, m = 64, 1024, 32
x = torch.randn(B, C, S, S, generator=g).to(dev)
w1 = (0.1 * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev); w2 = w1.clone()
gy = torch.randn(B, C, S, S, generator=g).to(dev)
y, xt = _native.spectral_conv2d_forward(x, w1, w2, S, S)
def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
tf = timed(lambda: _native.spectral_conv2d_forward(x, w1, w2, S, S))
tb = timed(lambda: _native.spectral_conv2d_backward(gy, xt, w1, w2, S, S))
img = B * C * S * S * 4; wb = 2 * C * C * m * m * 8
print(f"SpectralConv2d(64,64,1024,1024,32,32) B={B}: fwd {tf*1e6:.0f} us ({(2*img+wb)/tf/1e12:.2f} TB/s = {(2*img+wb)/tf/8e12*100:.1f} %)  bwd {tb*1e6:.0f} us ({(2*img+2*wb)/tb/8e12*100:.1f} %)")

xb, gyb = x.bfloat16(), gy.bfloat16()
yb, xtb = _native.spectral_conv2d_forward(xb, w1, w2, S, S)
tf = timed(lambda: _native.spectral_conv2d_forward(xb, w1, w2, S, S))
tb = timed(lambda: _native.spectral_conv2d_backward(gyb, xtb, w1, w2, S, S))
imgb = img // 2; wh = wb // 2          # s_a = 2 (bf16), s_w = 4 (complex half storage)
err = float((yb.float() - y).norm() / y.norm())
print(f"  mixed (bf16 activations) B={B}: fwd {tf*1e6:.0f} us ({(2*imgb+wh)/tf/1e12:.2f} TB/s = {(2*imgb+wh)/tf/8e12*100:.1f} % of 8 TB/s on {(2*imgb+wh)/1e6:.0f} MB)"
      f"  bwd {tb*1e6:.0f} us ({(2*imgb+2*wh)/tb/8e12*100:.1f} %)   |y_bf16 - y_f32| / |y_f32| = {err:.1e}")
