"""Tuning aid: the block-Jacobi ILU(0) set-up (extract, schedule, factor) on the bench matrix -- 100^3 TGV in lexicographic
atom order, the library's bricks -- REPS times; wall time per create.  For rocprofv3 --pmc runs (scripts/pmc_script.sh)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import isph_amd  # noqa: F401
from isph_amd import hip, workload

n = int(os.environ.get("ISPH_NCELL", "100"))
REPS = int(os.environ.get("ISPH_REPS", "4"))
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = hip.Context(0, stream=st.cuda_stream, ordering="bricks")
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
ts = []
for r in range(REPS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M = hip.Precond(ctx, A, os.environ.get("ISPH_PREC", "bjacobi-ilu0"), 0)
    ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
    M.close()
print("ilu set-up n=%d: ms per create %s" % (n, " ".join("%.2f" % t for t in ts)))
A.close(); ctx.close()
