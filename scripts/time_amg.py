"""Tuning aid: the SA-AMG preconditioner on the bench matrix (100^3 TGV, lexicographic atom order, the library's row
numbering): wall time of the hierarchy set-up alone (REPS creates) and of the FGMRES solve with a hierarchy at hand, and
the level sizes.  For rocprofv3 --kernel-trace runs (scripts/prof_amg.sh profiles the bench leg itself)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import isph_amd  # noqa: F401
from isph_amd import hip, workload

n = int(os.environ.get("ISPH_NCELL", "100"))
REPS = int(os.environ.get("ISPH_REPS", "5"))
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = hip.Context(0, stream=st.cuda_stream, ordering=os.environ.get("ISPH_ORDERING", "bricks"))
spec = workload.TGVSpec(dim=3, ncell=(n, n, n), brick=(n, n, n), mode=workload.ADVECT)
parts = workload.make_tgv(spec)
dp = dict(parts)
for k in ("x", "type", "neigh_ptr", "neigh_idx"):
    dp[k] = torch.from_numpy(np.ascontiguousarray(parts[k])).to(dev)
own = torch.from_numpy(parts["owner_index"].astype(np.int64)).to(dev)
colmap = own.to(torch.int32).contiguous()
rho = torch.from_numpy(parts["rho"]).to(dev)
vs = torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)
vf = hip.compute_volumes(ctx, dp, colmap)
A, b = hip.assemble_poisson(ctx, dp, colmap, spec.dt, rho, vs, vfrac=vf[own].contiguous())
N = n ** 3
nv = torch.full((N,), 1.0 / np.sqrt(float(N)), dtype=torch.float64, device=dev)
prm_amg = hip.AmgParams(block=int(os.environ.get("ISPH_BLOCK", "512")), theta=0.0)
ts = []
for r in range(REPS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M = hip.PrecondAMG(ctx, A, nullvec=nv, params=prm_amg)
    ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
    if r < REPS - 1:
        M.close()
print("sa-amg set-up n=%d: ms per create %s" % (n, " ".join("%.2f" % t for t in ts)))
print("levels:", [M.level_info(l) for l in range(M.levels)])
prm = hip.SolverParams(tol=1e-8)
x = torch.zeros(N, dtype=torch.float64, device=dev)
bw = b.clone()
ss = []
for r in range(REPS):
    bw.copy_(b); x.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inf = hip.solve(ctx, A, bw, x, prec=M, singular=True, params=prm)
    torch.cuda.synchronize()
    ss.append((time.perf_counter() - t0) * 1e3)
print("solve with the hierarchy at hand: ms %s, iterations %d" % (" ".join("%.2f" % t for t in ss), inf.iters))
M.close(); A.close(); ctx.close()
