"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/uno_spectral.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def libpath():
    from uno_amd import build
    return build.build()


def _declared():
    hdr = open(os.path.join(ROOT, "include", "uno_spectral.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(uno_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ("uno_spectral_conv2d_forward", "uno_spectral_conv2d_backward", "uno_dft2d_forward",
                 "uno_dft2d_inverse", "uno_mode_mix", "uno_mode_wgrad", "uno_last_error", "uno_abi_version"):
        assert must in names


def test_library_exports_every_declared_symbol(libpath):
    h = ctypes.CDLL(libpath)
    for name in _declared():
        assert hasattr(h, name), f"{name} declared in include/uno_spectral.h but not exported"


def test_binding_covers_header(libpath):
    from uno_amd import _native
    assert sorted(_native.EXPORTED_SYMBOLS) == _declared()
    lib = _native.lib()
    assert lib.uno_abi_version() == _native.ABI_VERSION
    assert lib.uno_spectral_conv2d_fwd_ws_bytes(2, 3, 4, 5, 6) == 8 * 2 * 4 * 2 * 5 * 6


def test_argument_errors_are_reported_without_a_gpu(libpath):
    """Argument validation happens before anything touches the device."""
    from uno_amd import _native
    lib = _native.lib()
    rc = lib.uno_dft2d_forward(None, None, 1, 8, 8, 2, 2, 1.0, 0, 0, None)
    assert rc < 0 and b"null" in lib.uno_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.uno_dft2d_forward(p, p, 1, 8, 8, 9, 2, 1.0, 0, 0, None)     # modes1 > rows
    assert rc < 0 and b"modes1" in lib.uno_last_error()
    rc = lib.uno_dft2d_forward(p, p, 1, 8, 8, 2, 6, 1.0, 0, 0, None)     # modes2 > cols/2+1
    assert rc < 0 and b"modes2" in lib.uno_last_error()


def test_argument_errors_of_the_block_level_entry_points(libpath):
    """Every entry point added for the rest of the operator block validates sizes / pointers on the host (no launch)."""
    from uno_amd import _native
    lib = _native.lib()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    nul = ctypes.c_void_p(0)
    cases = [
        (lib.uno_channel_mix(p, p, nul, p, 2, 0, 4, 10, 0, 0, 0, nul, None), b"bad sizes"),
        (lib.uno_channel_mix(nul, p, nul, p, 2, 3, 4, 10, 0, 0, 0, nul, None), b"null"),
        (lib.uno_channel_wgrad(p, p, nul, nul, p, 2, 3, 4, 10, 0, None), b"null"),
        (lib.uno_channel_wgrad(p, p, p, nul, p, 2, 3, -1, 10, 0, None), b"bad sizes"),
        (lib.uno_resample2d(p, p, p, 1, 0, 4, 4, 4, p, p, 1, p, p, 1, nul, nul, 0, 0, None), b"bad sizes"),
        (lib.uno_gelu_project_forward(p, p, nul, p, 1, 0, 5, None), b"bad sizes"),
        (lib.uno_gelu_project_backward(p, p, p, p, nul, nul, p, 1, 3, 5, None), b"null"),
        (lib.uno_gelu_pad(p, nul, p, 1, 4, 4, 3, 4, 0, None), b"bad sizes"),          # padded size smaller than the input
        (lib.uno_gelu_pad(p, nul, p, 1, 4, 4, 5, 5, 1, None), b"null"),               # backward without grad_output
        (lib.uno_instnorm_forward(p, nul, nul, p, p, p, 5, 2, 7, 1e-5, 1, None), b"bad sizes"),   # rows not a multiple of C
        (lib.uno_instnorm_backward(p, nul, nul, nul, p, p, p, p, p, 4, 2, 7, 1, None), b"null"),
        (lib.uno_adam_step(p, p, p, p, 4, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None), b"bad arguments"),   # step < 1
        (lib.uno_adam_step(p, p, p, p, 4, 0, 1e-3, 1.0, 0.999, 1e-8, 0.0, 1, None), b"bad arguments"),   # beta1 = 1
    ]
    for rc, needle in cases:
        assert rc < 0
    # messages are per call: check the last few individually
    assert lib.uno_gelu_pad(p, nul, p, 1, 4, 4, 3, 4, 0, None) < 0 and b"bad sizes" in lib.uno_last_error()
    assert lib.uno_channel_mix(nul, p, nul, p, 2, 3, 4, 10, 0, 0, 0, nul, None) < 0 and b"null" in lib.uno_last_error()
    # zero-sized problems are no-ops that succeed without touching the device
    assert lib.uno_channel_mix(nul, nul, nul, nul, 0, 3, 4, 10, 0, 0, 0, nul, None) == 0
    assert lib.uno_gelu_pad(nul, nul, nul, 0, 4, 4, 5, 5, 0, None) == 0
    assert lib.uno_instnorm_forward(nul, nul, nul, nul, nul, nul, 0, 2, 7, 1e-5, 1, None) == 0
    assert lib.uno_channel_wgrad_ws_bytes(2, 64, 64, 1000) > 0 and lib.uno_gelu_project_bwd_ws_bytes(2, 64, 5000) % (4 * 65) == 0 and lib.uno_gelu_project_bwd_ws_bytes(2, 64, 5000) > 0
