"""bench.py's weak-scaling decompositions (one 100^3 brick per rank on a 1x1x1 / 2x1x1 / 2x2x1 / 2x2x2 torus) with the ranks
as THREADS of one process on ONE GPU, over the host-staged transport (tests/ranks.py): not a performance figure -- the
ranks share the chip -- but the numerics of the N-rank runs the driver measures on N GPUs: iterations, residual, wall."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import isph_amd  # noqa: F401
from isph_amd import hip
import test_gpu_ranks as T
from ranks import RankGroup

hip.lib()
n = int(os.environ.get("ISPH_NCELL", "100"))
prec = os.environ.get("ISPH_PREC", "bjacobi-ilu0")   # or sa-amg
print("preconditioner", prec, flush=True)
for pgrid in ((1, 1, 1), (2, 1, 1), (2, 2, 1), (2, 2, 2)):
    world = int(np.prod(pgrid))
    G = RankGroup(world, timeout_s=600.0)
    try:
        res = G.run(T._config2_rank, n, pgrid, prec)
        cnt = G.counts()
    finally:
        G.close()
    N = float(sum(r["nl"] for r in res))
    rr = sum(r["rr"] for r in res) - sum(r["sum_r"] for r in res) ** 2 / N
    print("ranks %d grid %dx%dx%d  rows %9d  iterations %s  ||r - mean r|| / ||b|| %.2e  ghost columns per rank %d  wall %.2f s  exchanges %d all-reduces %d"
          % ((world,) + pgrid + (int(N), sorted({r["info"][1] for r in res}), np.sqrt(max(rr, 0) / sum(r["bb2"] for r in res)),
                                 res[0]["nghost"], max(r["wall"] for r in res), cnt["exchanges"], cnt["allreduces"])), flush=True)
