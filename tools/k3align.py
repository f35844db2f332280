# K3 time vs the byte offset of its output buffer (is the 230 / 278 us bimodality an alignment effect?)
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uno_amd import _native
B, Cc, S, m = 16, 64, 421, 20
dev = torch.device("cuda:0")
O = torch.randn(B, Cc, 2 * m, m, dtype=torch.cfloat, device=dev)
n = B * Cc * S * S
big = torch.empty(n + (1 << 22), device=dev)
L = _native.lib()
def run(off):
    out = big[off:off + n]
    def call():
        rc = L.uno_dft2d_inverse(C.c_void_p(O.data_ptr()), C.c_void_p(out.data_ptr()), B * Cc, S, S, m, m, C.c_float(1.0), 1, 1,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
print("base address % 2MiB =", big.data_ptr() % (1 << 21))
for off in [0, 1, 3, 16, 32, 64, 128, 256, 1024, 4096, 65536, 1 << 20]:
    print(f"offset {off:8d} floats: K3 {run(off):7.1f} us")
