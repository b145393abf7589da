"""Poisson assembly at 100^3 (lexicographic atom order), REPS times, for rocprofv3 --kernel-trace --stats
(scripts/prof_assembly.sh): library row order on / off by argv[1] = bricks | caller; wall time per assembly printed."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import isph_amd  # noqa: F401
from isph_amd import hip, workload

mode = sys.argv[1] if len(sys.argv) > 1 else "bricks"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
REPS = 6
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = hip.Context(0, stream=st.cuda_stream, ordering=mode)
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
vfrac = vf[own].contiguous()
ts = []
for r in range(REPS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    A, b = hip.assemble_poisson(ctx, dp, colmap, spec.dt, rho, vs, vfrac=vfrac)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    A.close()
print("assemble_poisson %s n=%d: ms per call %s" % (mode, n, " ".join("%.2f" % t for t in ts)))
ctx.close()
