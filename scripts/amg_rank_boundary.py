"""What rank-local ("Uncoupled") aggregates cost in iterations: the oracle on a 2-brick box (2N x N x N cells, the
bench's 2-GPU decomposition), FGMRES(50) + SA-AMG V cycle, tol 1e-8.
  full      hierarchy of the whole operator (aggregates may cross the rank boundary; ML with repartitioning off has none
            of that -- this is the one-rank reference)
  filtered  every entry coupling the two bricks dropped on EVERY level, the fine one included (lower bound of what a
            rank-local hierarchy can do: its smoother and residual miss the couplings too)
The device keeps the halo on the fine level (smoother + residual over A with ghost columns) and drops it above; on the
one-GPU self-peer plan its count sits at the `full` end (tests/test_gpu_multirank_shape.py: 14 / 14 / 23 and 12 / 13 / 23).
    python scripts/amg_rank_boundary.py [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import isph_amd  # noqa: E402,F401
from isph_amd import workload  # noqa: E402
import oracle as orc  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [12, 16]:
    spec = workload.TGVSpec(dim=3, ncell=(2 * n, n, n), brick=(4, 4, 4), mode=workload.JITTER)
    parts = workload.make_tgv(spec)
    P = orc.Particles(parts, workload.single_rank_colmap(parts)).precompute(corrections=False)
    rp, ci, val, b = P.poisson(spec.dt, parts["rho"], parts["v"], antisym=True)
    N = parts["nlocal"]
    rank = (parts["x"][:N, 0] >= parts["x"][:N, 0].max() / 2 + 1e-9).astype(np.int32)
    rows = np.repeat(np.arange(N), np.diff(rp))
    keep = rank[rows] == rank[ci]
    rpf = np.zeros(N + 1, np.int32)
    rpf[1:] = np.cumsum(np.bincount(rows[keep], minlength=N))
    nv = np.ones(N) / np.sqrt(N)
    out = {}
    for theta in (0.0, 0.02):
        kw = dict(theta=theta, block=256, coarse_max=64)
        Gf = orc.AMG(rp, ci, val, nullvec=nv, **kw)
        Gl = orc.AMG(rpf, ci[keep], val[keep], nullvec=nv, **kw)
        _, i_full, _ = orc.solve(rp, ci, val, b, singular=True, prec="amg", amg=Gf)
        _, i_filt, _ = orc.solve(rp, ci, val, b, singular=True, prec="amg", amg=Gl)
        out[theta] = (i_full.iters, i_filt.iters, Gf.levels, Gl.levels)
    print("2 bricks of %d^3 (%d rows, %.1f %% of the entries cross the rank boundary):" % (n, N, 100.0 * (1 - keep.mean())),
          "  ".join("theta %.2f: full %d / filtered %d iterations (levels %d / %d)" % ((t,) + out[t]) for t in out))
