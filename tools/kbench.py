"""Per-kernel micro-benchmark of the spectral block (developer tool; bench.py is the contract).
usage: python tools/kbench.py [B C S m] [iters]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native

args = [int(a) for a in sys.argv[1:]]
B, C, S, m = (args + [16, 64, 421, 20])[:4] if len(args) >= 4 else (16, 64, 421, 20)
iters = args[4] if len(args) > 4 else 20
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(B, C, S, S, generator=g).to(dev)
sc = (1 / (2 * C)) ** 0.5
w1 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
w2 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
gy = torch.randn(B, C, S, S, generator=g).to(dev)


def timeit(fn, n=iters, warm=3, reps=5):
    """median over `reps` timed groups of n launches (clock / box noise is +-10 % between single groups)"""
    for _ in range(warm):
        fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n)
    out.sort()
    return out[len(out) // 2]


img_bytes = B * C * S * S * 4
w_bytes = 2 * C * C * m * m * 8
X = _native.dft2d_forward(x, m, m, scale=1.0 / (S * S))
O = _native.mode_mix(X, [w1, w2], 0)
t1 = timeit(lambda: _native.dft2d_forward(x, m, m, scale=1.0 / (S * S)))
t2 = timeit(lambda: _native.mode_mix(X, [w1, w2], 0))
t3 = timeit(lambda: _native.dft2d_inverse(O, S, S))
t4 = timeit(lambda: _native.mode_mix(O, [w1, w2], 1))
t5 = timeit(lambda: _native.mode_wgrad(X, O, w1.shape, 2))
t6 = timeit(lambda: _native.dft2d_forward(gy, m, m, scale=1.0, hermitian_cols=True, mask_overlap=True))
t7 = timeit(lambda: _native.dft2d_inverse(O, S, S, scale=1.0 / (S * S), hermitian_cols=False, mask_overlap=False))
tf = timeit(lambda: _native.spectral_conv2d_forward(x, w1, w2, S, S))
y, xt = _native.spectral_conv2d_forward(x, w1, w2, S, S)
tb = timeit(lambda: _native.spectral_conv2d_backward(gy, xt, w1, w2, S, S))
tcopy = timeit(lambda: y.copy_(x))
fwd_bytes = 2 * img_bytes + w_bytes
bwd_bytes = 2 * img_bytes + 2 * w_bytes
print(f"config B={B} C={C} S={S} m={m}")
print(f"K1 dft fwd   {t1*1e3:8.1f} us  {img_bytes/t1/1e9:7.2f} TB/s")
print(f"K2 mix       {t2*1e3:8.1f} us")
print(f"K3 dft inv   {t3*1e3:8.1f} us  {img_bytes/t3/1e9:7.2f} TB/s")
print(f"K2' gX mix   {t4*1e3:8.1f} us")
print(f"K4 wgrad     {t5*1e3:8.1f} us")
print(f"K1 (bwd flags) {t6*1e3:6.1f} us   K3 (bwd flags) {t7*1e3:6.1f} us")
print(f"forward      {tf*1e3:8.1f} us  {fwd_bytes/tf/1e9:7.2f} TB/s  ({fwd_bytes/tf/1e9/8*100:.1f}% of 8 TB/s)")
print(f"backward     {tb*1e3:8.1f} us  {bwd_bytes/tb/1e9:7.2f} TB/s  ({bwd_bytes/tb/1e9/8*100:.1f}% of 8 TB/s)")
print(f"d2d copy     {tcopy*1e3:8.1f} us  {2*img_bytes/tcopy/1e9:7.2f} TB/s")
# stock PyTorch-ROCm comparator (rocFFT + rocBLAS)
def stock():
    xf = torch.fft.rfft2(x, norm="forward")
    out = torch.zeros(B, C, S, S // 2 + 1, dtype=torch.cfloat, device=dev)
    out[:, :, :m, :m] = torch.einsum("bixy,ioxy->boxy", xf[:, :, :m, :m], w1)
    out[:, :, -m:, :m] = torch.einsum("bixy,ioxy->boxy", xf[:, :, -m:, :m], w2)
    return torch.fft.irfft2(out, s=(S, S), norm="forward")
ts = timeit(stock, n=5, warm=2, reps=1)
print(f"stock torch fwd (rocFFT+rocBLAS) {ts*1e3:8.1f} us")
