"""The C-ABI library loads and exports every symbol include/tinybvh_amd.h declares; error
behaviour without a GPU is a status code, never exit()."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "tinybvh_amd.h")).read() + open(os.path.join(ROOT, "include", "tinybvh_amd_debug.h")).read()   # (the boundary + the development aids)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tbvh_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 30
    raw = C.CDLL(_capi.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in the header but not exported"
        assert s in _capi.SYMBOLS, f"{s} declared in the header but not bound in _capi.py"
    for s in _capi.SYMBOLS:
        assert s in syms, f"{s} bound but not declared in the header"


def test_abi_version_and_layout_constants():
    assert _capi.lib.tbvh_abi_version() == 5
    assert tb.RAY_DTYPE.itemsize == 64
    assert tb.RAY_DTYPE.fields["t"][1] == 48 and tb.RAY_DTYPE.fields["prim"][1] == 60
    assert tb.RAY_DTYPE.fields["inst"][1] == 44 and tb.RAY_DTYPE.fields["rD"][1] == 32
    assert C.sizeof(_capi.Camera) == 64 and C.sizeof(_capi.BuildParams) == 16


def test_errors_are_status_codes_not_exit():
    lib = _capi.lib
    h = C.c_void_p()
    # invalid arguments
    assert lib.tbvh_host_build(None, 10, tb.LAYOUT_CWBVH, None, C.byref(h)) == -1
    assert b"null" in lib.tbvh_last_error()
    v = np.zeros((3, 4), np.float32)
    assert lib.tbvh_host_build(C.c_void_p(v.ctypes.data), 1, 12345, None, C.byref(h)) == -1
    assert b"layout" in lib.tbvh_last_error()
    assert lib.tbvh_init(0, None) == -1
    assert lib.tbvh_intersect(None, None, 0, 64) == -1
    assert lib.tbvh_synchronize(None) == -1
    assert lib.tbvh_time_last_ms(None) == -1.0
    n = lib.tbvh_device_count()
    if n <= 0:
        # no GPU here: init must fail cleanly with NODEVICE
        assert lib.tbvh_init(0, C.byref(h)) in (-2, -3)
        assert lib.tbvh_last_error()
        with pytest.raises(tb.TbvhError):
            tb.Context(0)
    else:
        assert lib.tbvh_init(n + 5, C.byref(h)) == -2


def test_safercp_and_make_rays_follow_the_ray_constructor():
    x = np.array([0.0, -0.0, 1e-13, -1e-13, 2.0, -4.0], np.float32)
    r = tb.safercp(x)
    assert r[0] == np.float32(1e30) and r[2] == np.float32(1e30) and r[3] == np.float32(-1e30)
    assert r[4] == 0.5 and r[5] == -0.25
    rays = tb.make_rays([[1, 2, 3]], [[0, 0, 2]])
    assert rays["mask"][0] == 0xFFFF and rays["t"][0] == np.float32(1e30)
    assert np.allclose(rays["D"][0], [0, 0, 1]) and rays["rD"][0][2] == 1.0 and rays["rD"][0][0] == np.float32(1e30)
