"""VERDICT r3 item 3, measured on the ORACLE only (no GPU): does an intra-block row order shorten the dependency chains of
the block-Jacobi ILU(0) (levels per 512-row block bound k_ilu_solve_stream and k_ilu_factor) without costing FGMRES
iterations?  100^3 TGV system of bench.py (mode advect), blocks = the generator's bricks, orders applied as a symmetric
permutation inside every block (the preconditioner changes, the system does not).

  natural     the brick's own order (x fastest)                                    -- what ships
  mc27        27 colours (x mod 3, y mod 3, z mod 3), colour-major
  mc8         8 colours (parity of x, y, z), colour-major
  zebra_z     planes z even first, then z odd
  rb          red-black by parity of x+y+z
  diag        sorted by x+y+z (hyperplanes)
  rev_x       x reversed every other line (boustrophedon)
  brick AxBxC other brick shapes with the natural order

Output: levels (L sweep) per block -- mean / max -- and FGMRES(50) iterations to 1e-8 with the oracle."""
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path[:0] = [".", "oracle", "tests"]
import isph_amd  # noqa: E402,F401
from isph_amd import workload  # noqa: E402
import oracle as orc  # noqa: E402


def system(n, brick):
    spec = workload.TGVSpec(dim=3, ncell=(n, n, n), brick=brick, mode=workload.ADVECT)
    parts = workload.make_tgv(spec)
    P = orc.Particles(parts, workload.single_rank_colmap(parts)).precompute(corrections=False)
    rp, ci, val, b = P.poisson(spec.dt, parts["rho"], parts["v"], antisym=True)
    N = parts["nlocal"]
    tag0 = parts["tag"][:N].astype(np.int64) - 1
    xyz = np.stack([tag0 % n, (tag0 // n) % n, tag0 // (n * n)], axis=1)
    return sps.csr_matrix((val, ci, rp), shape=(N, N)), b, xyz


def levels_per_block(A, B, sample=200, seed=1):
    """L-sweep levels of ILU(0) on consecutive B-row blocks (in-block strictly-lower pattern), for a sample of blocks"""
    N = A.shape[0]
    nb = (N + B - 1) // B
    rng = np.random.default_rng(seed)
    pick = np.sort(rng.choice(nb, size=min(sample, nb), replace=False))
    out = []
    for bk in pick:
        lo, hi = bk * B, min(N, bk * B + B)
        S = A[lo:hi][:, lo:hi].tocsr()
        lev = np.zeros(hi - lo, dtype=np.int64)
        ip, ix = S.indptr, S.indices
        for i in range(hi - lo):
            c = ix[ip[i]:ip[i + 1]]
            c = c[c < i]
            lev[i] = 1 + (lev[c].max() if len(c) else 0)
        out.append(lev.max())
    return np.asarray(out)


def order_key(name, xyz, brick):
    l = xyz % np.asarray(brick)          # coordinates inside the brick
    x, y, z = l[:, 0], l[:, 1], l[:, 2]
    nat = x + brick[0] * (y + brick[1] * z)
    if name == "natural":
        return nat
    if name == "mc27":
        return ((x % 3) + 3 * (y % 3) + 9 * (z % 3)) * 100000 + nat
    if name == "mc8":
        return ((x % 2) + 2 * (y % 2) + 4 * (z % 2)) * 100000 + nat
    if name == "mc64":
        return ((x % 4) + 4 * (y % 4) + 16 * (z % 4)) * 100000 + nat
    if name == "zebra_z":
        return (z % 2) * 100000 + nat
    if name == "rb":
        return ((x + y + z) % 2) * 100000 + nat
    if name == "diag":
        return (x + y + z) * 100000 + nat
    if name == "rev_x":
        xr = np.where(y % 2 == 1, brick[0] - 1 - x, x)
        return xr + brick[0] * (y + brick[1] * z)
    raise ValueError(name)


def run(n, brick, orders, B=512):
    t0 = time.time()
    A, b, xyz = system(n, brick)
    N = A.shape[0]
    print("# %d^3, brick %s: system in %.1f s, %d entries" % (n, brick, time.time() - t0, A.nnz), flush=True)
    bp = np.arange(0, N + B, B).clip(0, N).astype(np.int32)
    blk = np.arange(N) // B
    rows = []
    for name in orders:
        key = order_key(name, xyz, brick)
        perm = np.lexsort((key, blk))                       # inside every block by key
        Ap = A[perm][:, perm].tocsr()
        Ap.sort_indices()
        lv = levels_per_block(Ap, B)
        t0 = time.time()
        ilu = orc.ILU(Ap.indptr, Ap.indices, Ap.data, 0, bp)
        x, info, _ = orc.solve(Ap.indptr, Ap.indices, Ap.data, b[perm], singular=True, prec="ilu", ilu=ilu)
        rows.append((name, lv.mean(), lv.max(), info.iters, info.converged, time.time() - t0))
        print("%-10s brick %-10s levels/block mean %6.1f max %4d   iterations %4d conv %d   (%.1f s)" %
              (name, "x".join(map(str, brick)), lv.mean(), lv.max(), info.iters, info.converged, time.time() - t0), flush=True)
    return rows


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    run(n, (8, 8, 8), ["natural", "mc27", "mc8", "mc64", "zebra_z", "rb", "diag", "rev_x"])
    if n % 16 == 0 or n == 100:
        pass
    for brick in ((4, 8, 16), (16, 8, 4), (4, 4, 32)):
        if all(n % k == 0 for k in brick):
            run(n, brick, ["natural"])
    # point Jacobi and no preconditioner on the same system, for scale
    A, b, xyz = system(n, (8, 8, 8))
    for pk in ("jacobi", "none"):
        x, info, _ = orc.solve(A.indptr, A.indices, A.data, b, singular=True, prec=pk)
        print("%-10s iterations %4d conv %d" % (pk, info.iters, info.converged), flush=True)
