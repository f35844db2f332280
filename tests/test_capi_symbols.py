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
