"""-m "not gpu": the C-ABI library builds for gfx950, loads, and exports every
symbol include/isph_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

import isph_amd
from isph_amd import build, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(isph_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def hiplib():
    path = build.build_hip()
    return ctypes.CDLL(path)


def test_every_declared_symbol_is_exported(hiplib):
    names = _declared("isph_hip.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(hiplib, n), "missing export: " + n
    assert sorted(hip.EXPORTS) == names


def test_host_library_exports():
    lib = ctypes.CDLL(build.build_host())
    for n in _declared("isph_workload.h"):
        assert hasattr(lib, n)


def test_no_gpu_means_loud_failure_not_fallback(hiplib):
    """Without a device the context cannot be created: the product never
    computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    hiplib.isph_ctx_create.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = hiplib.isph_ctx_create(0, None, ctypes.byref(h))
    assert rc == -1
    hiplib.isph_last_error.restype = ctypes.c_char_p
    assert b"no HIP device" in hiplib.isph_last_error() or hiplib.isph_last_error()


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "implicit-sph_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "isph_oracle" not in txt and "import oracle" not in txt, f
