"""Tuning aid: device time of one block-Jacobi ILU(0) apply on the bench matrix (HIP events via torch on the ctx stream)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import isph_amd
from isph_amd import hip, workload, dist
import ctypes as C
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
torch.cuda.synchronize(); t0 = time.perf_counter()
M = hip.Precond(ctx, A, "bjacobi-ilu0", int(os.environ.get("ISPH_BLOCK", "512"))); ctx.sync()
print("ilu_create %.2f ms" % ((time.perf_counter() - t0) * 1e3), "info", M.info())
n = parts["nlocal"]
r = torch.randn(n, dtype=torch.float64, device=dev); z = torch.zeros_like(r)
L = hip.lib()
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(50): L.isph_prec_apply(ctx.h, M.h, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr()), 1)
    ctx.sync(); t1 = time.perf_counter()
    print("prefetch %s: apply %.4f ms" % (os.environ.get("ISPH_ILU_PREFETCH", "8"), (t1 - t0) * 1e3 / 50))
print("checksum %.12e" % float(z.double().abs().sum()))
