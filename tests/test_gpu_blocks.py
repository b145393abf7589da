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


def test_caller_subdomains_are_validated(gpu_ctx):
    pr = Problem(tgv_spec(dim=3, n=12))
    rp, ci, val, b = pr.poisson()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    n = pr.n
    for bad in ([0, 100, 100, n], [0, 1100, n], [1, n], [0, n - 1]):
        with pytest.raises(hip.IsphError):
            hip.Precond(gpu_ctx, A, "bjacobi-ilu0", block_ptr=np.asarray(bad, dtype=np.int32))
    A.close()
