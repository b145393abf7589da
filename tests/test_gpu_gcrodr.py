"""-m gpu: "Recycling GMRES" (SolverLin_Belos "Solver Type", solver_lin_belos.h:178-179 -> Belos::GCRODRSolMgr) on the
device (csrc/gcrodr.hpp) against the oracle's numpy restatement of GCRO-DR(m, k) (oracle/gcrodr.py): iterations +-1,
pressure vector 1e-6, and the recycling pays against restarted GMRES with the same m."""
import numpy as np
import pytest

from isph_amd import hip, workload
import oracle as orc
import gcrodr as gcro
from problems import Problem, tgv_spec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,k,prec", [(10, 4, "none"), (20, 5, "bjacobi-ilu0"), (12, 6, "jacobi"), (50, 20, "none")])
def test_gcrodr_matches_oracle(gpu_ctx, m, k, prec):
    pr = Problem(tgv_spec(dim=3, n=16, mode=workload.JITTER))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    M = hip.Precond(gpu_ctx, A, prec, 256)
    if prec == "bjacobi-ilu0":
        bp = np.arange(0, n + 256, 256).clip(0, n).astype(np.int32)
        minv = orc.ILU(rp, ci, val, 0, bp).apply
    elif prec == "jacobi":
        d = np.array([val[rp[i]:rp[i + 1]][ci[rp[i]:rp[i + 1]] == i][0] for i in range(n)])
        minv = lambda r: r / d
    else:
        minv = None
    xo, io = gcro.solve(rp, ci, val, b, singular=True, prec=minv, num_blocks=m, num_recycled=k)
    x = np.zeros(n)
    info = hip.solve(gpu_ctx, A, b.copy(), x, prec=M, singular=True,
                     params=hip.SolverParams(solver_type=2, num_blocks=m, num_recycled=k))
    assert info.converged == 1 and io["converged"]
    assert abs(info.iters - io["iters"]) <= 1 and info.restarts == io["restarts"]
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
    assert abs(x.mean()) < 1e-12 * np.abs(x).max()
    # against restarted GMRES(m) (non-flexible, same preconditioner): recycling never needs more iterations here
    xg = np.zeros(n)
    ig = hip.solve(gpu_ctx, A, b.copy(), xg, prec=M, singular=True, params=hip.SolverParams(num_blocks=m, flexible=0))
    assert ig.converged == 1 and info.iters <= ig.iters + 1
    assert np.linalg.norm(x - xg) <= 1e-6 * np.linalg.norm(xg)


def test_gcrodr_rejects_reference_default_list(gpu_ctx):
    """"Num Recycled Blocks" = 50 with "Num Blocks" = 50 (the reference's default list, solver_lin_belos.h:226-240) is
    invalid for GCRO-DR -- Belos::GCRODRSolMgr throws; here the call fails with a message, no fallback."""
    pr = Problem(tgv_spec(dim=2, n=16, mode=workload.JITTER))
    rp, ci, val, b = pr.poisson()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    x = np.zeros(pr.n)
    with pytest.raises(hip.IsphError, match="Num Recycled Blocks"):
        hip.solve(gpu_ctx, A, b.copy(), x, singular=True, params=hip.SolverParams(solver_type=2))
