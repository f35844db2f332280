"""3-D spectral convolution on the HIP path (SpectralConv3d_Uno.forward, reference
integral_operators.py:385-427)."""
from __future__ import annotations


def spectral_conv3d(x, weights, dim1, dim2, dim3):
    raise NotImplementedError("uno_amd: the 3-D HIP spectral convolution is not built yet")
