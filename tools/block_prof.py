"""Standalone spectral blocks for rocprofv3, run with THE SAME warm protocol as bench.py's `_timed` (3 untimed groups, then
5 groups of 20 back-to-back calls), forward then backward:
    c2: SpectralConv2d(64,64,421,421,20,20), batch 16      (BASELINE.json's roofline block; SURVEY 8(d): fwd 1478.2 MB, bwd 1504.4 MB)
    c4: SpectralConv3d(32,32,64,64,20,16,16,8), batch 8
usage: python tools/block_prof.py c2|c4 [calls per group = 20] [timed groups = 5] [warm groups = 3]
       (run under rocprofv3 --kernel-trace [--stats] / --pmc ...; tools/block_rocprof_summary.py reads the per-dispatch trace)

Segments are separated by a marker launch (uno::gelu_pad_fwd_kernel on a 4-element tensor - a kernel the blocks never use):
    marker | warm forward | marker | timed forward | marker | warm backward | marker | timed backward | marker
The host-side HIP-event time of the timed groups is printed as well, so that one run yields both readings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
GROUPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
WARM = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
tiny = torch.zeros(1, 1, 2, 2, device=dev)


def marker():
    _native.gelu_pad(tiny, 2, 2)


if which == "c2":
    B, C, S, m = 16, 64, 421, 20
    x = torch.randn(B, C, S, S, generator=g).to(dev)
    sc = (1 / (2 * C)) ** 0.5
    w1 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
    w2 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
    gy = torch.randn(B, C, S, S, generator=g).to(dev)
    y, xt = _native.spectral_conv2d_forward(x, w1, w2, S, S)
    fwd = lambda: _native.spectral_conv2d_forward(x, w1, w2, S, S)
    bwd = lambda: _native.spectral_conv2d_backward(gy, xt, w1, w2, S, S)
else:
    B, C, H, W, T, m1, m2, m3 = 8, 32, 64, 64, 20, 16, 16, 8
    x = torch.randn(B, C, H, W, T, generator=g).to(dev)
    sc = (1 / (2 * C)) ** 0.5
    ws = [(sc * torch.randn(C, C, m1, m2, m3, dtype=torch.cfloat, generator=g)).to(dev) for _ in range(4)]
    gy = torch.randn(B, C, H, W, T, generator=g).to(dev)
    y, xt = _native.spectral_conv3d_forward(x, ws, H, W, T)
    fwd = lambda: _native.spectral_conv3d_forward(x, ws, H, W, T)
    bwd = lambda: _native.spectral_conv3d_backward(gy, xt, ws, H, W, T)
bwd()
torch.cuda.synchronize()
for name, fn in (("forward", fwd), ("backward", bwd)):
    marker()
    for _ in range(WARM * N):
        fn()
    torch.cuda.synchronize()
    marker()
    times = []
    for _ in range(GROUPS):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / N * 1e3)
    times.sort()
    print(f"{which} {name}: HIP-event us per call, groups of {N}: {[round(t, 1) for t in times]} median {times[len(times) // 2]:.1f}")
marker()
torch.cuda.synchronize()
print("done", which, N, GROUPS, WARM)
