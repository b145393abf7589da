"""BASELINE configs[4]-shaped run at a chosen size (bcc lattice, Quintic cut 3h, MorrisHolmes, NotSingular Poisson,
FGMRES + SA-AMG), stage by stage with timings.  usage: python scripts/run_config4.py <cells per side> [prec]
nc = 114 -> 2.96 M particles / 2.2e9 matrix entries (64-bit offsets), nc = 126 -> 4.0 M / 3.0e9."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import isph_amd  # noqa: F401
from isph_amd import hip, workload

nc = int(sys.argv[1])
prec = sys.argv[2] if len(sys.argv) > 2 else "sa-amg"
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
T0 = time.time()


def stage(msg):
    torch.cuda.synchronize()
    ctx.sync()
    used = (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9
    pi = hip.pool_info(reset_peak=True)
    print("[%7.1fs] %s | device memory in use %.1f GB; library: live %.1f GB, peak live since the last line %.1f GB, cached %.1f GB" %
          (time.time() - T0, msg, used, pi["live"] / 1e9, pi["peak_live"] / 1e9, pi["cached"] / 1e9), flush=True)


if os.environ.get("ISPH_REFERENCE_BEADS"):   # the script's own bead pack and aspect ratio (tests/golden/...npz)
    p = workload.make_porous_cylinder(nc, bead_pack=np.load(os.path.join(ROOT, "tests", "golden", "pore_scale_flow_bead_centeroids_3d.npz")))
else:
    p = workload.make_porous_cylinder(nc, nbeads=40, rbead_cells=6.0)
n, nall = p["nlocal"], p["nall"]
stage("generated n=%d list entries=%d offsets %s" % (n, int(p["neigh_ptr"][-1]), p["neigh_ptr"].dtype))
colmap_h = workload.single_rank_colmap(p)
dp = dict(p)
for k in ("x", "type", "neigh_ptr", "neigh_idx"):
    dp[k] = torch.from_numpy(np.ascontiguousarray(p[k])).to(dev)
colmap = torch.from_numpy(colmap_h).to(dev)
own = torch.from_numpy(p["owner_index"].astype(np.int64)).to(dev)
rho = torch.from_numpy(p["rho"]).to(dev)
stage("particles on the device")
vf = hip.compute_volumes(ctx, dp, colmap, kernel="quintic")
vfrac = vf[own].contiguous()
pnd = (1.0 / vfrac).contiguous()
stage("volumes")
x = dp["x"]
fluid = (dp["type"] <= 2).to(torch.float64)
vstar = torch.zeros((nall, 3), dtype=torch.float64, device=dev)
vstar[:, 1] = 1e-3 * torch.cos(x[:, 0]) * fluid
vstar[:, 0] = 1e-3 * torch.sin(x[:, 1]) * fluid
A, b = hip.assemble_poisson(ctx, dp, colmap, p["dt"], rho, vstar, singular=hip.NOT_SINGULAR, vfrac=vfrac,
                            kernel="quintic", kinds=p["kinds"], pnd=pnd)
im = A.info()
stage("assembled nnz=%d (%.1f per row), SELL %.1f GB" % (im["nnz"], im["nnz"] / n, im["sell_bytes"] / 1e9))
del dp["neigh_idx"], dp["neigh_ptr"]
p.pop("neigh_idx")
torch.cuda.empty_cache()
y = A.spmv(torch.ones(n, dtype=torch.float64, device=dev))
stage("SpMV ok, max |A 1| on fluid rows %.3e" % float(y[dp["type"][:n] <= 2].abs().max()))
for rnd in ("first", "steady"):  # the reference rebuilds the preconditioner every solve: the second round is what a time step sees
    t0 = time.time()
    if prec == "sa-amg":
        M = hip.PrecondAMG(ctx, A, params=hip.AmgParams(block=512))
        stage("%s: AMG hierarchy %s  (%.2f s)" % (rnd, [M.level_info(l)["rows"] for l in range(M.levels)], time.time() - t0))
    else:
        M = hip.Precond(ctx, A, prec, 512)
        stage("%s: %s built (%.2f s)" % (rnd, prec, time.time() - t0))
    xs = torch.zeros(n, dtype=torch.float64, device=dev)
    bw = b.clone()
    t0 = time.time()
    info = hip.solve(ctx, A, bw, xs, prec=M, singular=False)
    stage("%s: solve converged=%d iterations=%d  %.2f s" % (rnd, info.converged, info.iters, time.time() - t0))
    M.close()
    del M
r = b - A.spmv(xs)
solid = dp["type"][:n] >= 3
print("residual %.3e, |x| on solid rows %.3e of %.3e" % (float(r.norm() / b.norm()), float(xs[solid].abs().max()), float(xs.abs().max())), flush=True)
