"""Regenerates the committed fixtures of this directory.

  reference_known_answers.json  numbers recorded by the reference itself: the 2-D Taylor-Green convergence table
                                sph-script/conv-taylor-green-vortex-2d-rev390.txt:6-34 and the one functor output
                                quoted in SURVEY.md Appendix A.  Written by hand from those files; this script only
                                checks that it is present.
  tgv2d_walls_12.npz            seeded 2-D TGV box (12x12 cells, jittered) with a solid slab: inputs are rebuilt from
                                the seed by tests; stored are the EXPECTED outputs of the CPU oracle at the time of
                                writing -- Poisson matrix (both operator families), right-hand side, ILU(0) factor,
                                GMRES solution, Helmholtz matrix and right-hand side, AMG aggregates.
The oracle restates the reference (oracle/isph_oracle.h says how far it is pinned); the npz pins the oracle AND the
GPU path against silent drift.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def golden_problem():
    import isph_amd  # noqa: F401
    from isph_amd import workload
    import oracle as orc
    from problems import Problem, tgv_spec, wall_types
    kinds = [orc.FLUID, orc.SOLID]
    out = {}
    for fam, antisym in (("antisym", True), ("sym", False)):
        pr = Problem(tgv_spec(dim=2, n=12, mode=workload.JITTER, brick=4), antisym=antisym, singular=orc.NOT_SINGULAR,
                     kinds=kinds, types=wall_types)
        out[fam] = pr
    return out


def build():
    import oracle as orc
    prs = golden_problem()
    data = {}
    for fam, pr in prs.items():
        rp, ci, val, b = pr.poisson()
        data[fam + "_rowptr"], data[fam + "_colidx"], data[fam + "_val"], data[fam + "_b"] = rp, ci, val, b
        p = pr.parts
        nall = p["nall"]
        x = p["x"]
        pres = np.cos(x[:, 0]) * np.sin(x[:, 1])
        force = np.ascontiguousarray(0.01 * np.stack([np.sin(x[:, 1]), np.cos(x[:, 0]), np.zeros(nall)], axis=1))
        g = np.array([0.05, -0.02, 0.0])
        rph, cih, vh, bh = pr.P.helmholtz(pr.spec.dt, 0.5, p["nu"], p["rho"], pres, force, g, np.ascontiguousarray(p["v"]),
                                          antisym=pr.antisym)
        data[fam + "_helm_val"], data[fam + "_helm_b"] = vh, bh
    pr = prs["antisym"]
    rp, ci, val, b = pr.poisson()
    bp = np.arange(0, pr.n + 64, 64).clip(0, pr.n).astype(np.int32)
    ilu = orc.ILU(rp, ci, val, 0, bp)
    frp, fci, fv = ilu.export()
    data["ilu_val"] = fv
    f1rp, f1ci, f1v = orc.ILU(rp, ci, val, 1, bp).export()      # "fact: level-of-fill" = 1, the reference's default
    data["ilu1_rowptr"], data["ilu1_colidx"], data["ilu1_val"] = f1rp, f1ci, f1v
    xs, info, _ = orc.solve(rp, ci, val, b, singular=False, prec="ilu", ilu=ilu)
    data["x"], data["iters"] = xs, np.array([info.iters])
    amg = orc.AMG(rp, ci, val, theta=0.05, block=64, coarse_max=16)
    data["amg_aggregates"] = amg.aggregates(0)
    np.savez_compressed(os.path.join(HERE, "tgv2d_walls_12.npz"), **data)
    json.load(open(os.path.join(HERE, "reference_known_answers.json")))
    print("wrote", os.path.join(HERE, "tgv2d_walls_12.npz"), {k: np.asarray(v).shape for k, v in data.items()})


if __name__ == "__main__":
    build()
