"""Error of a split-bf16 row stage for FLOAT32 images: x = x0 + x1 + x2, twiddle = t0 + t1 + t2 (bf16 pieces, round to
nearest even), products accumulated in float32 as the bf16 MFMA does - 6 products (all terms down to 2^-16) and 3 products (down to 2^-8)
- against the float64 truncated DFT, next to the plain float32 transform (what K1-HT computes).  CPU only (numpy emulation: a bf16 x bf16
product is exact in float32; the accumulation order differs from the MFMA's, the error level does not).
    python tools/dev/split_error.py > profiles/r04_split_bf16_error.txt"""
import numpy as np

def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.view(np.float32)

def split3(x):
    x = x.astype(np.float32)
    a = bf16(x); r = x - a
    b = bf16(r); r2 = r - b
    c = bf16(r2)
    return a, b, c

def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))

rng = np.random.default_rng(0)
print("row stage T[h, l] = sum_w x[h, w] exp(-2 pi i l w / W): relative L2 error against float64")
print(f"{'shape (rows x W, modes)':28s} {'float32 (K1-HT)':>16s} {'6 products':>12s} {'3 products':>12s} {'hi+lo x, hi+lo t (3)':>22s}")
for (R, W, m2, dist) in [(64, 421, 20, "randn"), (64, 421, 20, "uniform[0,1)"), (64, 446, 18, "randn"), (64, 223, 18, "randn"), (64, 111, 8, "randn"),
                         (64, 1024, 32, "randn"), (16, 85, 12, "randn"), (21, 18, 5, "randn"), (23, 23, 4, "randn")]:
    x = (rng.standard_normal((R, W)) if dist == "randn" else rng.random((R, W))).astype(np.float32)
    w = np.arange(W)[:, None] * np.arange(m2)[None, :]
    ang = 2 * np.pi * (w % W) / W
    tw = np.concatenate([np.cos(ang), -np.sin(ang)], axis=1)                # (W, 2 m2) float64
    exact = x.astype(np.float64) @ tw
    f32 = (x @ tw.astype(np.float32)).astype(np.float64)
    x0, x1, x2 = split3(x)
    t0, t1, t2 = split3(tw.astype(np.float32))
    six = (x0 @ t0 + (x0 @ t1 + x1 @ t0) + (x1 @ t1 + x0 @ t2 + x2 @ t0)).astype(np.float64)
    three = (x0 @ t0 + (x0 @ t1 + x1 @ t0)).astype(np.float64)
    two2 = ((x0 @ t0) + (x0 @ t1) + (x1 @ t0)).astype(np.float64)
    print(f"{str((R, W, m2)) + ' ' + dist:28s} {rel(f32, exact):16.2e} {rel(six, exact):12.2e} {rel(three, exact):12.2e} {rel(two2, exact):22.2e}")
print()
print("full forward transform of a 421 x 421 float32 image, 20 x 20 modes (C2 block layer), column stage in float32:")
H = W = 421; m1 = m2 = 20
x = rng.standard_normal((H, W)).astype(np.float32)
rows = np.concatenate([np.arange(m1), np.arange(H - m1, H)])
Fh = np.exp(-2j * np.pi * ((rows[:, None] * np.arange(H)[None, :]) % H) / H)
Fw = np.exp(-2j * np.pi * ((np.arange(W)[:, None] * np.arange(m2)[None, :]) % W) / W)
exact = Fh @ (x.astype(np.float64) @ Fw)
def col(T):
    return (Fh.astype(np.complex64) @ T.astype(np.complex64)).astype(np.complex128)
twr, twi = Fw.real.astype(np.float32), Fw.imag.astype(np.float32)
f32 = col((x @ twr) + 1j * (x @ twi))
x0, x1, x2 = split3(x)
def six(t):
    t0, t1, t2 = split3(t)
    return x0 @ t0 + (x0 @ t1 + x1 @ t0) + (x1 @ t1 + x0 @ t2 + x2 @ t0)
def three(t):
    t0, t1, t2 = split3(t)
    return x0 @ t0 + (x0 @ t1 + x1 @ t0)
print(f"   float32 row stage {rel(f32, exact):.2e}   6 products {rel(col(six(twr) + 1j * six(twi)), exact):.2e}   3 products {rel(col(three(twr) + 1j * three(twi)), exact):.2e}")
print()
print("bar set by VERDICT r3: adopt only if <= 2e-6 (parity tolerance TOL = 2e-5 untouched).")
