"""The path `north_star` names, unchanged: PairISPH hands SolverLin_Belos a HOST Epetra CSR
(ref: pair_isph.cpp:924-926,988-1011 -> solver_lin_belos.h:130-222).  Here: the pipelined host ingress of
isph_mat_create_csr (csrc/ingress.hpp) and SolverLin_HIP::solveProblem on BASELINE configs[1] at its own size."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
from problems import tgv_spec  # noqa: E402
from isph_amd import hip, workload  # noqa: E402

pytestmark = pytest.mark.gpu


def _rand_csr(rng, n, ncol, maxlen, sort):
    lens = rng.integers(0, maxlen + 1, size=n)
    rp = np.zeros(n + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    ci = np.empty(rp[-1], np.int32)
    for i in range(n):
        c = rng.choice(ncol, size=lens[i], replace=False)
        ci[rp[i]:rp[i + 1]] = np.sort(c) if sort else c
    val = rng.standard_normal(rp[-1])
    return rp, ci, val


@pytest.mark.parametrize("n,maxlen,sort", [(1, 3, True), (63, 9, True), (64, 9, False), (1000, 40, True), (1000, 40, False),
                                           (70000, 33, True)])
def test_host_csr_ingress_round_trip(gpu_ctx, n, maxlen, sort):
    """Host CSR -> sliced-ELL through the pipelined ingress -> export: the same matrix (rows come back column-sorted,
    duplicates never occur here), ragged rows, empty rows, sizes around a slice and beyond one ring chunk (70 000 x 33
    entries > 2^20), sorted rows (no row sort) and unsorted ones (the device raises the flag and sorts)."""
    rng = np.random.default_rng(n + maxlen)
    ncol = n + 17
    rp, ci, val = _rand_csr(rng, n, ncol, maxlen, sort)
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val, ncol=ncol)
    rp2, ci2, v2 = A.export_csr()
    assert np.array_equal(rp2, rp)
    for i in range(0, n, max(1, n // 997)):
        o = np.argsort(ci[rp[i]:rp[i + 1]], kind="stable")
        assert np.array_equal(ci2[rp[i]:rp[i + 1]], ci[rp[i]:rp[i + 1]][o])
        assert np.array_equal(v2[rp[i]:rp[i + 1]], val[rp[i]:rp[i + 1]][o])
    x = rng.standard_normal(ncol)
    y = A.spmv(x)
    import scipy.sparse as sps
    yr = sps.csr_matrix((val, ci, rp), shape=(n, ncol)) @ x
    assert np.allclose(y, yr, rtol=0, atol=1e-12 * max(1.0, np.abs(yr).max()))
    A.close()


@pytest.mark.parametrize("n,band,sort", [(3000, 20000, True), (200000, 30000, True), (200000, 30000, False), (50000, 400000, True)])
def test_host_csr_ingress_packs_columns_where_the_rows_allow_it(gpu_ctx, n, band, sort):
    """Rows whose sorted neighbours lie less than 65536 columns apart cross the link as 16-bit differences (10 bytes
    per entry instead of 12, the first column of every row in a table): the matrix must come out the same, and the
    ingress must report the smaller transfer.  Unsorted rows and rows with wider gaps go as 32-bit columns."""
    rng = np.random.default_rng(band)
    ncol = n + band
    lens = rng.integers(1, 40, size=n)
    rp = np.zeros(n + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    ci = np.empty(rp[-1], np.int32)
    for i in range(n):
        c = i + rng.choice(band, size=lens[i], replace=False)
        ci[rp[i]:rp[i + 1]] = np.sort(c) if sort else c
    val = rng.standard_normal(rp[-1])
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val, ncol=ncol)
    info = hip.ingress_info(gpu_ctx)
    nnz = int(rp[-1])
    packed = sort and band <= 65536
    if packed:
        assert info["link_bytes"] < 10.5 * nnz + 8 * n + 64, info
    else:
        assert info["link_bytes"] >= 12 * nnz, info
    rp2, ci2, v2 = A.export_csr()
    assert np.array_equal(rp2, rp)
    if sort:
        assert np.array_equal(ci2, ci) and np.array_equal(v2, val)
    else:
        for i in range(0, n, 997):
            o = np.argsort(ci[rp[i]:rp[i + 1]], kind="stable")
            assert np.array_equal(ci2[rp[i]:rp[i + 1]], ci[rp[i]:rp[i + 1]][o]) and np.array_equal(v2[rp[i]:rp[i + 1]], val[rp[i]:rp[i + 1]][o])
    A.close()


def test_host_csr_ingress_refuses_bad_operands(gpu_ctx):
    rp = np.array([0, 2, 4], np.int32)
    ci = np.array([0, 1, 0, 5], np.int32)          # column 5 of a 2-column matrix, found by the copy pass
    v = np.ones(4)
    with pytest.raises(hip.IsphError, match="column index out of range"):
        hip.Matrix.from_csr(gpu_ctx, rp, ci, v, ncol=2)
    with pytest.raises(hip.IsphError, match="column index out of range"):
        hip.Matrix.from_csr(gpu_ctx, rp, np.array([0, 1, -1, 1], np.int32), v, ncol=2)
    with pytest.raises(hip.IsphError, match="rowptr not monotone"):
        hip.Matrix.from_csr(gpu_ctx, np.array([0, 3, 2], np.int32), ci[:3], v[:3], ncol=6)
    # and the context is still usable afterwards
    A = hip.Matrix.from_csr(gpu_ctx, rp, np.array([0, 1, 0, 1], np.int32), v, ncol=2)
    assert np.allclose(A.spmv(np.array([1.0, 2.0])), [3.0, 3.0])
    A.close()


def test_dropin_solver_lin_at_config1_size(gpu_ctx, tmp_path):
    """BASELINE configs[1] (3-D TGV, 100^3 = 10^6 rows, 104 M entries) through the UNCHANGED drop-in: the system is
    exported to the host and handed to the C++ mirror exactly as PairISPH would (host CSR, host b, host x) --
    SolverLin_Belos::solveProblem + PrecondWrapper_Ifpack (fill 0, 512-row subdomains).  Must give the iteration count
    and the pressure vector of the device-assembly path (<= 1e-6, the tolerance north_star's parity is stated at; the
    two runs factor the same matrix, so they agree far below that), and the timed mode must report a sane split."""
    from isph_amd import build
    exe = build.build_cpp_test()
    sp = tgv_spec(dim=3, n=100, mode=workload.ADVECT)
    p = workload.make_tgv(sp)
    colmap = workload.single_rank_colmap(p)
    n = p["nlocal"]
    vf = hip.compute_volumes(gpu_ctx, p, colmap)
    vfrac = np.ascontiguousarray(vf[p["owner_index"]])
    A, b = hip.assemble_poisson(gpu_ctx, p, colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]), vfrac=vfrac)
    M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 512)
    x, bb = np.zeros(n), b.copy()
    info = hip.solve(gpu_ctx, A, bb, x, prec=M, singular=True)
    assert info.converged == 1
    rp, ci, v = A.export_csr()
    M.close(); A.close()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    fin, fout = os.path.join(base, "isph_dropin_sys.bin"), os.path.join(base, "isph_dropin_x.bin")
    try:
        with open(fin, "wb") as f:
            np.array([n, len(v)], np.int32).tofile(f)
            rp.astype(np.int32).tofile(f); ci.astype(np.int32).tofile(f); v.tofile(f); b.tofile(f)
        r = subprocess.run([exe, fin, fout, "1", "timed", "3"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out = np.fromfile(fout)
    finally:
        for f_ in (fin, fout):
            if os.path.exists(f_):
                os.remove(f_)
    xd, bproj = out[:n], out[n:]
    rec = [json.loads(l)["dropin"] for l in r.stdout.splitlines() if l.startswith('{"dropin"')][0]
    assert rec["converged"] == 1 and rec["iterations"] == info.iters, (rec, info.iters)
    assert np.linalg.norm(xd - x) / np.linalg.norm(x) <= 1e-6
    assert np.allclose(bproj, bb, rtol=0, atol=1e-13 * np.abs(bb).max())      # b view projected in place like the reference
    assert rec["rows"] == n and rec["entries"] == len(v)
    assert 0 < rec["ingress_ms"] < rec["ms_per_solve"] and rec["krylov_ms"] > 0
    print("dropin:", rec)


@pytest.mark.parametrize("n,block,shuffle,brick", [(16, 512, False, 8), (40, 512, False, 8), (40, 256, False, 8), (24, 512, True, 8),
                                                   (40, 512, False, 1)])
def test_fused_ingress_equals_separate_calls(gpu_ctx, n, block, shuffle, brick):
    """isph_mat_create_csr_bjacobi (ILU(0) set-up queued range by range behind the arriving rows) against
    isph_mat_create_csr + isph_prec_create: the same matrix, the same factor and the same application, bit for bit.
    40^3 = 64 000 rows x 104 entries is more than one 4 Mi-entry chunk and several set-up batches; `shuffle` hands over
    rows that are not column-sorted (the fused set-up is then redone on the sorted image); brick 1 = lexicographic
    particle order, whose rows cross the link as 16-bit column differences."""
    sp = tgv_spec(dim=3, n=n, mode=workload.JITTER, brick=brick)
    p = workload.make_tgv(sp)
    colmap = workload.single_rank_colmap(p)
    vf = hip.compute_volumes(gpu_ctx, p, colmap)
    A0, b = hip.assemble_poisson(gpu_ctx, p, colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]),
                                 vfrac=np.ascontiguousarray(vf[p["owner_index"]]))
    rp, ci, v = A0.export_csr()
    A0.close()
    nrow = len(rp) - 1
    if shuffle:
        rng = np.random.default_rng(5)
        ci, v = ci.copy(), v.copy()
        for i in range(0, nrow, 7):
            o = rng.permutation(rp[i + 1] - rp[i])
            ci[rp[i]:rp[i + 1]] = ci[rp[i]:rp[i + 1]][o]
            v[rp[i]:rp[i + 1]] = v[rp[i]:rp[i + 1]][o]
    A1 = hip.Matrix.from_csr(gpu_ctx, rp, ci, v)
    M1 = hip.Precond(gpu_ctx, A1, "bjacobi-ilu0", block)
    A2, M2 = hip.Matrix.from_host_csr_with_bjacobi(gpu_ctx, rp, ci, v, block)
    for a, c in zip(A1.export_csr(), A2.export_csr()):
        assert np.array_equal(a, c)
    for a, c in zip(M1.export_ilu(), M2.export_ilu()):
        assert np.array_equal(a, c)
    assert M1.info() == M2.info()
    r = np.random.default_rng(1).standard_normal(nrow)
    assert np.array_equal(M1.apply(r), M2.apply(r))
    assert np.array_equal(A1.spmv(r), A2.spmv(r))
    x1, x2 = np.zeros(nrow), np.zeros(nrow)
    i1 = hip.solve(gpu_ctx, A1, b.copy(), x1, prec=M1, singular=True)
    i2 = hip.solve(gpu_ctx, A2, b.copy(), x2, prec=M2, singular=True)
    assert i1.iters == i2.iters and np.array_equal(x1, x2)
    for o in (M1, M2, A1, A2):
        o.close()


def test_host_csr_ingress_degenerate_shapes(gpu_ctx):
    """No entries at all, rows without entries at both ends, one dense row: the pipeline's chunk / slice book-keeping at
    its edges (no chunk, a first chunk that is also the last, slices that complete without a copy)."""
    n = 130
    A = hip.Matrix.from_csr(gpu_ctx, np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0))
    assert A.info()["nnz"] == 0 and np.array_equal(A.spmv(np.ones(n)), np.zeros(n))
    A.close()
    rp = np.zeros(n + 1, np.int32)
    rp[70:] = n                                          # row 69 is dense, everything else is empty
    ci = np.arange(n, dtype=np.int32)
    v = np.arange(1.0, n + 1.0)
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, v)
    y = A.spmv(np.ones(n))
    assert y[69] == v.sum() and np.count_nonzero(y) == 1
    rp2, ci2, v2 = A.export_csr()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(v2, v)
    A.close()
    # the fused entry on an identity matrix: every block factors trivially, M^-1 = I
    m = 1000
    Af, Mf = hip.Matrix.from_host_csr_with_bjacobi(gpu_ctx, np.arange(m + 1, dtype=np.int32), np.arange(m, dtype=np.int32), np.full(m, 2.0), 64)
    r = np.linspace(-1.0, 1.0, m)
    assert np.array_equal(Mf.apply(r), r / 2.0) and np.array_equal(Af.spmv(r), 2.0 * r)
    Mf.close(); Af.close()
