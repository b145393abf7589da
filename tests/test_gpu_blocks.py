"""-m gpu: block-Jacobi ILU(0) on the CALLER'S subdomains (isph_prec_create_blocks): row ranges of any length up to 1024
that need not start on a 64-row slice.  The reference's subdomains are the bricks of the spatial decomposition
(precond_ifpack.h:60-74: one Ifpack subdomain per rank, rows = the rank's particles, pair_isph.cpp:1258-1259); the
oracle factors the same row ranges (orc.ILU block_ptr).  Factor pattern exact, values 1e-10, application 1e-11,
FGMRES iterations +-1."""
import numpy as np
import pytest

from isph_amd import hip, workload
import oracle as orc
from problems import Problem, tgv_spec

pytestmark = pytest.mark.gpu


def _tables(n, kind):
    if kind == "bricks500":                      # 10 x 10 x 5 bricks of the generator
        return np.arange(0, n + 500, 500).clip(0, n).astype(np.int32)
    if kind == "ragged":                         # anything from 1 to 1024 rows, no alignment
        rng = np.random.default_rng(11)
        cuts, at = [0], 0
        while at < n:
            at = min(n, at + int(rng.choice([1, 7, 37, 64, 100, 333, 500, 512, 777, 1000, 1024])))
            cuts.append(at)
        return np.asarray(cuts, dtype=np.int32)
    if kind == "uniform512":
        return np.arange(0, n + 512, 512).clip(0, n).astype(np.int32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["bricks500", "ragged", "uniform512"])
def test_ilu0_on_caller_subdomains_matches_oracle(gpu_ctx, kind):
    spec = workload.TGVSpec(dim=3, ncell=(20, 20, 20), brick=(10, 10, 5), mode=workload.JITTER)
    pr = Problem(spec)
    rp, ci, val, b = pr.poisson()
    n = pr.n
    bp = _tables(n, kind)
    ref = orc.ILU(rp, ci, val, 0, bp)
    frp, fci, fv = ref.export()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", block_ptr=bp)
    grp, gci, gv = M.export_ilu()
    assert np.array_equal(grp, frp) and np.array_equal(gci, fci)
    assert np.max(np.abs(gv - fv) / np.maximum(np.abs(fv), 1e-300 + 1e-10 * np.abs(fv).max())) < 1e-10
    r = np.random.default_rng(5).standard_normal(n)
    z, zo = M.apply(r), ref.apply(r)
    assert np.linalg.norm(z - zo) / np.linalg.norm(zo) < 1e-11
    if kind == "uniform512":                     # the table form of the built-in decomposition: the same object, bit for bit
        M2 = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 512)
        assert np.array_equal(M2.apply(r), z)
        M2.close()
    xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="ilu", ilu=ref)
    bg, xg = b.copy(), np.zeros(n)
    info = hip.solve(gpu_ctx, A, bg, xg, prec=M, singular=True)
    assert info.converged == 1 and abs(info.iters - io.iters) <= 1, (info.iters, io.iters)
    assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) <= 1e-6
    M.close(); A.close()


@pytest.mark.parametrize("kind,fill", [("bricks500", 1), ("ragged", 1), ("ragged", 2), ("uniform512", 1)])
def test_iluk_on_caller_subdomains_matches_oracle(gpu_ctx, kind, fill):
    """"fact: level-of-fill" k > 0 (precond_ifpack.h:35; the reference's default is 1) on the caller's subdomains of any
    length: the level-of-fill pattern of every block exact against orc.ILU(rp, ci, val, k, bp) (Ifpack_IlukGraph's rule
    restated), values 1e-10, application 1e-11, FGMRES iterations +-1; the table form of uniform blocks is the built-in
    "bjacobi-ilu<k>" object bit for bit."""
    spec = workload.TGVSpec(dim=3, ncell=(20, 20, 20), brick=(10, 10, 5), mode=workload.JITTER)
    pr = Problem(spec)
    rp, ci, val, b = pr.poisson()
    n = pr.n
    bp = _tables(n, kind)
    ref = orc.ILU(rp, ci, val, fill, bp)
    frp, fci, fv = ref.export()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    M = hip.Precond(gpu_ctx, A, "bjacobi-ilu%d" % fill, block_ptr=bp)
    grp, gci, gv = M.export_ilu()
    assert np.array_equal(grp, frp) and np.array_equal(gci, fci)
    assert len(fci) > len(ci) * 0.5                                        # the pattern did fill in
    assert np.max(np.abs(gv - fv) / np.maximum(np.abs(fv), 1e-10 * np.abs(fv).max())) < 1e-10
    r = np.random.default_rng(5).standard_normal(n)
    z, zo = M.apply(r), ref.apply(r)
    assert np.linalg.norm(z - zo) / np.linalg.norm(zo) < 1e-11
    if kind == "uniform512":
        M2 = hip.Precond(gpu_ctx, A, "bjacobi-ilu%d" % fill, 512)
        assert np.array_equal(M2.apply(r), z)
        M2.close()
    xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="ilu", ilu=ref)
    bg, xg = b.copy(), np.zeros(n)
    info = hip.solve(gpu_ctx, A, bg, xg, prec=M, singular=True)
    assert info.converged == 1 and abs(info.iters - io.iters) <= 1, (info.iters, io.iters)
    assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) <= 1e-6
    M.close(); A.close()


def test_tables_of_tiny_blocks_do_not_overrun_the_factor_arrays(gpu_ctx):
    """a table of 1-row and few-row blocks on a banded matrix: every block's factor region is rounded up to 64 entries,
    so the regions together exceed the matrix' stored entries -- the arrays are sized for that; factor and
    application against the oracle (1-row blocks: M = diag(A))"""
    n = 3000
    main = 4.0 + np.arange(n) * 1e-3
    import scipy.sparse as sps
    T = sps.diags([-1.0 * np.ones(n - 1), main, -1.2 * np.ones(n - 1)], [-1, 0, 1], format="csr")
    rp, ci, val = T.indptr.astype(np.int32), T.indices.astype(np.int32), T.data.copy()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    for bp in (np.arange(n + 1, dtype=np.int32), np.arange(0, n + 10, 10).clip(0, n).astype(np.int32),
               np.unique(np.r_[0, np.arange(1, n, 3), n]).astype(np.int32)):
        ref = orc.ILU(rp, ci, val, 0, bp)
        M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", block_ptr=bp)
        f1, f2 = M.export_ilu(), ref.export()
        assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])
        assert np.max(np.abs(f1[2] - f2[2])) <= 1e-12 * np.abs(f2[2]).max()
        r = np.random.default_rng(1).standard_normal(n)
        assert np.linalg.norm(M.apply(r) - ref.apply(r)) <= 1e-12 * np.linalg.norm(r)
        M.close()
    A.close()


def test_caller_subdomains_are_validated(gpu_ctx):
    pr = Problem(tgv_spec(dim=3, n=12))
    rp, ci, val, b = pr.poisson()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    n = pr.n
    for bad in ([0, 100, 100, n], [0, 1100, n], [1, n], [0, n - 1]):
        with pytest.raises(hip.IsphError):
            hip.Precond(gpu_ctx, A, "bjacobi-ilu0", block_ptr=np.asarray(bad, dtype=np.int32))
    A.close()


def test_fused_host_ingress_with_caller_subdomains_is_the_two_step_result(gpu_ctx):
    """isph_mat_create_csr_blocks (host CSR crossing PCIe with the set-up of the caller's subdomains running behind the
    rows that have arrived) == isph_mat_create_csr + isph_prec_create_blocks, bit for bit."""
    spec = workload.TGVSpec(dim=3, ncell=(30, 30, 30), brick=(10, 10, 5), mode=workload.JITTER)
    pr = Problem(spec)
    rp, ci, val, b = pr.poisson()
    n = pr.n
    for kind in ("bricks500", "ragged"):
        bp = _tables(n, kind)
        A1 = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
        M1 = hip.Precond(gpu_ctx, A1, "bjacobi-ilu0", block_ptr=bp)
        A2, M2 = hip.Matrix.from_host_csr_with_bjacobi(gpu_ctx, rp, ci, val, block_ptr=bp)
        r = np.random.default_rng(3).standard_normal(n)
        assert np.array_equal(M1.apply(r), M2.apply(r))
        f1, f2 = M1.export_ilu(), M2.export_ilu()
        assert all(np.array_equal(a, c) for a, c in zip(f1, f2))
        assert np.array_equal(A1.spmv(r), A2.spmv(r))
        for o in (M1, M2, A1, A2):
            o.close()


def test_cpp_mirror_with_subdomain_table(gpu_ctx, tmp_path):
    """PrecondWrapper_Ifpack::setSubdomains through SolverLin_Belos::solveProblem (fused ingress with the table)."""
    import subprocess
    from isph_amd import build
    spec = workload.TGVSpec(dim=3, ncell=(20, 20, 20), brick=(10, 10, 5), mode=workload.JITTER)
    pr = Problem(spec)
    rp, ci, val, b = pr.poisson()
    n = pr.n
    exe = build.build_cpp_test()
    fin, fout = tmp_path / "sys.bin", tmp_path / "x.bin"
    with open(fin, "wb") as f:
        np.array([n, len(val)], np.int32).tofile(f)
        rp.astype(np.int32).tofile(f); ci.astype(np.int32).tofile(f); val.tofile(f); b.tofile(f)
    r = subprocess.run([exe, str(fin), str(fout), "1", "timed", "2", "500"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    x = np.fromfile(fout)[:n]
    bp = _tables(n, "bricks500")
    xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="ilu", ilu=orc.ILU(rp, ci, val, 0, bp))
    its = [int(t.split(":")[1].split(",")[0]) for t in r.stdout.split('"iterations"')[1:2]]
    assert abs(its[0] - io.iters) <= 1, (its, io.iters)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6


def test_profile_mode_counts_the_launches_of_every_kernel_class(gpu_ctx):
    """isph_ctx_profile_read: HIP events around the hot kernels by class.  One set-up + one FGMRES solve with the block
    ILU(0): one extract / schedule / factor launch, one preconditioner application and one Gram-Schmidt step (three
    sweeps) per iteration, SpMVs = iterations + one residual per cycle + the explicit residual; the solve's own
    spmv_calls agrees; times are positive and add up to less than the wall time of the bracket."""
    import time
    pr = Problem(tgv_spec(dim=3, n=24, mode=workload.JITTER))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    gpu_ctx.sync()
    gpu_ctx.set_profile(True)
    try:
        t0 = time.perf_counter()
        M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 512)
        x, bb = np.zeros(n), b.copy()
        info = hip.solve(gpu_ctx, A, bb, x, prec=M, singular=True)
        prof = gpu_ctx.profile_read()
        wall_ms = (time.perf_counter() - t0) * 1e3
    finally:
        gpu_ctx.set_profile(False)
    cycles = info.restarts + 1
    assert prof["ilu_extract"][1] == prof["ilu_schedule"][1] == prof["ilu_factor"][1] == 1
    assert prof["prec_apply"][1] in (info.iters, info.iters + 1)          # + 1: a speculative application behind the last column
    assert prof["multi_dot"][1] == prof["multi_axpy_dot"][1] == prof["multi_axpy_norm"][1] == info.iters
    assert prof["spmv"][1] in (info.iters + cycles + 1, info.iters + cycles + 2) and info.spmv_calls == prof["spmv"][1]
    assert all(ms > 0.0 for ms, calls in prof.values() if calls) and sum(ms for ms, _ in prof.values()) < wall_ms
    assert gpu_ctx.profile_read()["spmv"] == (0.0, 0)                     # reading starts a new collection
    M.close(); A.close()
