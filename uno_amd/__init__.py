"""uno_amd - MI355X-native spectral-convolution hot path of U-NO (ashiq24/UNO).

Package layout (only what the path needs):
  csrc/                  hand-written gfx950 kernels + the C ABI (include/uno_spectral.h)
  build.py               hipcc build of lib/libuno_spectral.so
  _native.py             ctypes binding of the C ABI
  integral_operators.py  host-side mirror of the reference's operator-block interface
  harness/               own counterparts of the reference callers (UNO_9, Adam, LpLoss, DDP step)
"""
__version__ = "0.1.0"
