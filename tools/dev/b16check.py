"""Development check of the bf16-MFMA transforms: relative L2 error of K1-B / K3-B against the float64 oracle per shape (no assert)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from uno_amd import _native
from oracle import spectral_oracle as so
dev = torch.device('cuda:0')
def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))
shapes = [(2, 64, 257, 8, 17), (2, 64, 257, 8, 16), (2, 64, 256, 8, 17), (2, 64, 288, 8, 8), (1, 16, 512, 4, 4), (1, 33, 1089, 16, 32),
          (3, 272, 272, 8, 8), (2, 100, 544, 18, 18), (1, 128, 1024, 32, 32), (5, 48, 300, 40, 16), (4, 50, 96, 20, 33), (2, 19, 700, 8, 48)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (n, H, W, m1, m2) in shapes:
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(n, 1, H, W, generator=g).bfloat16()
    got = _native.dft2d_forward(x.to(dev), m1, m2).cpu().numpy()
    want = so.truncated_rfft2_dense(x.double().numpy(), m1, m2) * (H * W)
    e = rel(got, want)
    msg = f"{(n, H, W, m1, m2)}: fwd {e:.2e}"
    if e > 2e-5:
        per_l = [rel(got[..., l], want[..., l]) for l in range(m2)]
        per_j = [rel(got[..., j, :], want[..., j, :]) for j in range(2 * m1)]
        msg += "\n   per mode l: " + " ".join(f"{v:.0e}" for v in per_l) + "\n   per row j: " + " ".join(f"{v:.0e}" for v in per_j)
    O = torch.randn(n, 1, 2 * m1, m2, dtype=torch.cfloat, generator=g)
    y = _native.dft2d_inverse(O.to(dev), H, W, dtype=torch.bfloat16).float().cpu().numpy()
    yw = so.truncated_irfft2_dense(O.numpy().astype(np.complex128), H, W, m1, m2)
    e2 = rel(y, yw)
    msg += f"   inv {e2:.2e}"
    if e2 > 3e-3:
        per_w = [rel(y[..., w0:w0 + 16], yw[..., w0:w0 + 16]) for w0 in range(0, W, 16)]
        per_h = [rel(y[..., h0:h0 + 16, :], yw[..., h0:h0 + 16, :]) for h0 in range(0, H, 16)]
        msg += "\n   per 16 columns: " + " ".join(f"{v:.0e}" for v in per_w) + "\n   per 16 rows: " + " ".join(f"{v:.0e}" for v in per_h)
    print(msg, flush=True)
