"""Tuning aid: create / apply time of the additive-Schwarz path on the bench matrix (ISPH_BLOCK rows per subdomain,
ISPH_OVERLAP layers, ISPH_COMBINE add|zero, ISPH_LEVEL_LAUNCHES=1 forces a launch per level)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import isph_amd
from isph_amd import hip, workload, dist
dev = torch.device("cuda", 0)
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
nc = int(os.environ.get("ISPH_NCELL", "100"))
spec = workload.TGVSpec(dim=3, ncell=(nc, nc, nc), brick=(8, 8, 8), mode=workload.ADVECT)
parts = workload.make_tgv(spec)
plan = dist.make_plan(parts, None)
dp = dict(parts)
for k in ("x", "type", "neigh_ptr", "neigh_idx"): dp[k] = torch.from_numpy(np.ascontiguousarray(parts[k])).to(dev)
colmap = torch.from_numpy(plan.colmap).to(dev); rho = torch.from_numpy(parts["rho"]).to(dev)
vstar = torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)
own = torch.from_numpy(parts["owner_index"].astype(np.int64)).to(dev)
vf = hip.compute_volumes(ctx, dp, colmap); vfrac = vf[own].contiguous()
A, b = hip.assemble_poisson(ctx, dp, colmap, spec.dt, rho, vstar, vfrac=vfrac, ncol=plan.ncol)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    M = hip.PrecondSchwarz(ctx, A, level_of_fill=int(os.environ.get("ISPH_FILL", "0")), overlap=int(os.environ.get("ISPH_OVERLAP", "1")),
                           combine=os.environ.get("ISPH_COMBINE", "zero"), block_size=int(os.environ.get("ISPH_BLOCK", "512")),
                           level_launches=bool(int(os.environ.get("ISPH_LEVEL_LAUNCHES", "0"))))
    ctx.sync()
    print("create %.1f ms" % ((time.perf_counter() - t0) * 1e3), M.schwarz_info(), M.create_timing())
    if rep < 2: M.close()
n = parts["nlocal"]
r = torch.randn(n, dtype=torch.float64, device=dev); z = torch.zeros_like(r)
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(10): M.apply(r, z)
    ctx.sync()
    print("apply %.3f ms" % ((time.perf_counter() - t0) * 1e3 / 10))
