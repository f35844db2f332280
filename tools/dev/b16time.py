import torch, sys, os
sys.path.insert(0, '.')
from uno_amd import _native
if os.environ.get('UNO_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['UNO_LIB'])
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(256, 1024, 1024, 32, 32), (256, 1089, 1089, 18, 18), (1024, 446, 446, 18, 18), (1024, 421, 421, 20, 20), (2048, 272, 272, 8, 8)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
which = os.environ.get("B16_WHICH", "fi")
for (n, H, W, m1, m2) in shapes:
    xs = [torch.randn(n, 1, H, W, device=dev).bfloat16() for _ in range(3)]
    O = torch.randn(n, 1, 2 * m1, m2, dtype=torch.cfloat, device=dev)
    i = [0]
    def f():
        i[0] += 1
        return _native.dft2d_forward(xs[i[0] % 3], m1, m2)
    def g():
        return _native.dft2d_inverse(O, H, W, dtype=torch.bfloat16)
    by = n * H * W * 2
    msg = f"{n}x{H}x{W} m=({m1},{m2}):"
    if "f" in which:
        tf = timeit(f); msg += f" fwd {tf:.1f} us = {by/tf/1e6:.2f} TB/s"
    if "i" in which:
        ti = timeit(g); msg += f"   inv {ti:.1f} us = {by/ti/1e6:.2f} TB/s"
    print(msg, flush=True)
