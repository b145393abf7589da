"""Iteration-count gap between the device's AMG variant (MIS-2 roots, block-local Gauss-Seidel) and ML's own definition
(sequential Uncoupled sweep, processor-wide Gauss-Seidel), measured with the CPU oracle on the 3-D TGV Poisson system.
TEST INFRASTRUCTURE: runs in the build container, prints the table DESIGN.md quotes."""
import sys
import os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), ".."), os.path.join(os.path.dirname(__file__), "..", "oracle"),
                os.path.join(os.path.dirname(__file__), "..", "tests")]
import numpy as np
import oracle as orc
from isph_amd import workload
from problems import Problem, tgv_spec

for n in [int(a) for a in sys.argv[1:]] or [32, 48]:
    pr = Problem(tgv_spec(dim=3, n=n, mode=workload.ADVECT))
    rp, ci, val, b = pr.poisson()
    nv = np.full(pr.n, 1.0 / np.sqrt(pr.n))
    for theta in (0.0, 0.02):
        row = []
        for agg, whole in (("mis2", False), ("mis2", True), ("ml", False), ("ml", True)):
            G = orc.AMG(rp, ci, val, nullvec=nv, theta=theta, aggregation=agg, whole_sgs=whole)
            x, info, _ = orc.solve(rp, ci, val, b, singular=True, prec="amg", amg=G)
            lv = [G.level_info(l)["rows"] for l in range(G.levels)]
            row.append("%s/%s: %d its %s" % (agg, "whole" if whole else "block", info.iters, lv))
        print("n=%d^3 theta=%.2f | " % (n, theta) + " | ".join(row), flush=True)
