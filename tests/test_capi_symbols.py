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
    for n in _declared("isph_workload.h") + _declared("isph_lammps.h"):
        assert hasattr(lib, n)


def test_lammps_format_converters():
    """include/isph_lammps.h: LAMMPS' full neighbour list (ilist / numneigh / int** firstneigh with special-bond bits in
    the top two bits of every index, functor.h:83-86, `jlist[jj] & NEIGHMASK`) -> the CSR of isph_particles, and the
    tag -> matrix-column map Epetra builds in FillComplete (functor_graph.h:61-97).  Index work: exact."""
    import numpy as np
    C = ctypes
    lib = C.CDLL(build.build_host())
    lib.isph_flatten_neighbor_list.restype = C.c_longlong
    rng = np.random.default_rng(3)
    nlocal, nall = 7, 11
    rows = [rng.choice(nall, size=rng.integers(0, 6), replace=False).astype(np.int32) for _ in range(nall)]
    bits = [((rng.integers(0, 4, size=len(r)).astype(np.int64) << 30) | r).astype(np.uint32).view(np.int32) for r in rows]
    numneigh = np.array([len(r) for r in rows], dtype=np.int32)
    first = (C.POINTER(C.c_int) * nall)(*[b.ctypes.data_as(C.POINTER(C.c_int)) for b in bits])
    ilist = rng.permutation(nlocal).astype(np.int32)                        # LAMMPS lists the owned atoms in bin order
    ptr, ptr64 = np.zeros(nlocal + 1, np.int32), np.zeros(nlocal + 1, np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    tot = lib.isph_flatten_neighbor_list(nlocal, p(ilist), p(numneigh), first, nlocal, nall, p(ptr), p(ptr64), None)
    assert tot == int(numneigh[:nlocal].sum()) and np.array_equal(ptr64, np.concatenate([[0], np.cumsum(numneigh[:nlocal])]))
    idx = np.zeros(tot, np.int32)
    assert lib.isph_flatten_neighbor_list(nlocal, p(ilist), p(numneigh), first, nlocal, nall, p(ptr), p(ptr64), p(idx)) == tot
    assert np.array_equal(ptr, ptr64)
    for i in range(nlocal):
        assert np.array_equal(idx[ptr[i]:ptr[i + 1]], rows[i])             # the bond bits are gone, the order is kept
    bad = ilist.copy()
    bad[1] = bad[0]
    assert lib.isph_flatten_neighbor_list(nlocal, p(bad), p(numneigh), first, nlocal, nall, p(ptr), p(ptr64), None) == -1
    # tags: owned 10..16, ghosts: a periodic image of an owned atom, two remote atoms, one of them seen twice
    tag = np.array([10, 11, 12, 13, 14, 15, 16, 12, 40, 41, 40], dtype=np.int32)
    colmap, gt = np.zeros(nall, np.int32), np.zeros(nall, np.int32)
    ncol = lib.isph_colmap_from_tags(nlocal, nall, p(tag), p(colmap), p(gt))
    assert ncol == 9 and list(colmap) == [0, 1, 2, 3, 4, 5, 6, 2, 7, 8, 7] and list(gt[:2]) == [40, 41]
    tag[3] = 10
    assert lib.isph_colmap_from_tags(nlocal, nall, p(tag), p(colmap), None) == -1   # two owned atoms with one tag


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


def test_python_plumbing_refuses_operands_shorter_than_the_abi_reads():
    """The C ABI takes bare pointers; a per-particle array shorter than what the entry point reads ([nall] fields,
    [nlocal] rows of Gc / Lc) would be read past its end.  The ctypes layer checks the element counts before the call
    (no device needed: the check comes first)."""
    import numpy as np
    import isph_amd  # noqa: F401
    from isph_amd import hip, workload
    spec = workload.TGVSpec(dim=2, ncell=(8, 8), brick=(4, 4), origin=(0.5, 0.5), mode=workload.JITTER)
    p = workload.make_tgv(spec)
    n, nall = p["nlocal"], p["nall"]
    assert nall > n
    colmap = workload.single_rank_colmap(p)
    ok = dict(vfrac=np.ones(nall), Gc=np.zeros((n, 4)), Lc=np.zeros((n, 3)))
    hip.particles_view(p, colmap, **ok)                                    # the sizes the ABI documents
    for bad in (dict(ok, vfrac=np.ones(n)), dict(ok, Gc=np.zeros((n - 1, 4))), dict(ok, Lc=np.zeros((n, 2))),
                dict(ok, pnd=np.ones(n)), dict(ok, normal=np.zeros((n, 3)))):
        with pytest.raises(ValueError):
            hip.particles_view(p, colmap, **bad)
    with pytest.raises(ValueError):
        hip.particles_view(p, colmap[:n], **ok)
    with pytest.raises(ValueError):                                        # rho given for the owned particles only
        hip.assemble_poisson(None, p, colmap, spec.dt, np.ones(n), np.zeros((nall, 3)), vfrac=np.ones(nall))
