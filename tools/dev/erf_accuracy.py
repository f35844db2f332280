"""Accuracy of the library's branch-free float32 erf (csrc/uno_common.h: uno_erf, the rational approximant evaluated here with the
same float32 operation order) and of the GELU built on it, against float64 - next to torch's own float32 F.gelu."""
import numpy as np
import torch
from scipy.special import erf

A = [-2.72614225801306e-10, 2.77068142495902e-08, -2.10102402082508e-06, -5.69250639462346e-05, -7.34990630326855e-04,
     -2.95459980854025e-03, -1.60960333262415e-02]
B = [-1.45660718464996e-05, -2.13374055278905e-04, -1.68282697438203e-03, -7.37332916720468e-03, -1.42647390514189e-02]


def uno_erf(x):
    x = np.clip(x.astype(np.float32), -4, 4)
    t = x * x
    p = np.float32(A[0])
    for c in A[1:]:
        p = p * t + np.float32(c)
    p = p * x
    q = np.float32(B[0])
    for c in B[1:]:
        q = q * t + np.float32(c)
    return (p / q).astype(np.float32)


x = np.linspace(-6, 6, 2000001).astype(np.float32)
t = erf(x.astype(np.float64))
e = uno_erf(x)
print("erf : max abs err %.3e  max rel err %.3e" % (np.abs(e - t).max(), (np.abs(e - t) / np.maximum(np.abs(t), 1e-30)).max()))
xn = (np.random.default_rng(0).standard_normal(2000000) * 2).astype(np.float32)
gt = 0.5 * xn.astype(np.float64) * (1 + erf(xn.astype(np.float64) / np.sqrt(2)))
g = 0.5 * xn * (1 + uno_erf(xn * np.float32(0.70710678118654752440)))
tt = torch.nn.functional.gelu(torch.from_numpy(xn)).numpy()
rl2 = lambda a: np.linalg.norm(a - gt) / np.linalg.norm(gt)
print("gelu: library rel L2 %.3e max abs %.3e | torch float32 rel L2 %.3e max abs %.3e" % (rl2(g), np.abs(g - gt).max(), rl2(tt), np.abs(tt - gt).max()))
