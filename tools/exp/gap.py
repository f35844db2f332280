import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
import bench
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, S, m = 16, 64, 421, 20
x = torch.randn(B, C, S, S, generator=g).to(dev)
sc = (1 / (2 * C)) ** 0.5
w1 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
w2 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
fwd = lambda: _native.spectral_conv2d_forward(x, w1, w2, S, S)
fwd()
for iters, reps in ((10, 3), (10, 3), (20, 5), (50, 3)):
    print("timed", iters, reps, f"{bench._timed(fwd, dev, iters=iters, reps=reps)*1e6:.1f} us")
# host enqueue cost of one call
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): fwd()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/50:.1f} us/call, total {1e6*(t2-t0)/50:.1f} us/call")
# preallocated outputs via raw C call
L = _native.lib()
import ctypes as Ct
y = torch.empty((B, C, S, S), device=dev); xt = torch.empty((B, C, 2*m, m), dtype=torch.complex64, device=dev)
ws = torch.empty(L.uno_spectral_conv2d_fwd_ws_bytes(B, C, C, m, m), dtype=torch.uint8, device=dev)
st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
raw = lambda: L.uno_spectral_conv2d_forward(Ct.c_void_p(x.data_ptr()), Ct.c_void_p(w1.data_ptr()), Ct.c_void_p(w2.data_ptr()), Ct.c_void_p(y.data_ptr()), Ct.c_void_p(xt.data_ptr()), Ct.c_void_p(ws.data_ptr()), B, C, C, S, S, S, S, m, m, st)
raw()
print("raw C call, fixed buffers", f"{bench._timed(raw, dev, iters=20, reps=5)*1e6:.1f} us")
