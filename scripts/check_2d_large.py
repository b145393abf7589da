"""2-D Taylor-Green lattice at scale (2048^2 = 4.2 M rows x 25 entries): assembly + solves with the block ILU(0) and the
SA-AMG preconditioner, host-side residual check.  usage on the GPU box: python scripts/check_2d_large.py [ncell]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import isph_amd  # noqa: F401
from isph_amd import hip, workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = hip.Context(0, stream=st.cuda_stream)
spec = workload.TGVSpec(dim=2, ncell=(n, n), brick=(16, 16), origin=(0.5, 0.5), mode=workload.JITTER)
t0 = time.perf_counter()
parts = workload.make_tgv(spec)
print("generated %d particles (%d with ghosts) in %.1f s" % (parts["nlocal"], parts["nall"], time.perf_counter() - t0), flush=True)
colmap = workload.single_rank_colmap(parts)
N = parts["nlocal"]
vf = hip.compute_volumes(ctx, parts, colmap)
vfrac = vf[parts["owner_index"]]
for prec in ("bjacobi-ilu0", "sa-amg"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    A, b = hip.assemble_poisson(ctx, parts, colmap, spec.dt, parts["rho"], np.ascontiguousarray(parts["v"]), vfrac=vfrac)
    t1 = time.perf_counter()
    if prec == "sa-amg":
        M = hip.PrecondAMG(ctx, A, nullvec=np.full(N, 1.0 / np.sqrt(N)))
    else:
        M = hip.Precond(ctx, A, prec, 512)
    x = np.zeros(N)
    bb = b.copy()
    info = hip.solve(ctx, A, bb, x, prec=M, singular=True)
    t2 = time.perf_counter()
    r = bb - A.spmv(x)
    r -= r.mean()
    rel = np.linalg.norm(r) / np.linalg.norm(bb)
    print("%s: rows %d nnz %d | assemble (host arrays in) %.0f ms | set-up + solve %.0f ms | its %d conv %d | host-side residual %.2e | x.1 %.1e"
          % (prec, N, A.info()["nnz"], (t1 - t0) * 1e3, (t2 - t1) * 1e3, info.iters, info.converged, rel, abs(x.sum()) / np.abs(x).sum()), flush=True)
    # restarted GMRES(50) with a one-level preconditioner stalls on a 2-D Poisson problem of this size (500 iterations are
    # Belos' limit, non-convergence is reported, not raised); the multigrid preconditioner must converge
    assert rel <= (2e-8 if prec == "sa-amg" else 1e-5) and (info.converged == 1 or prec != "sa-amg")
    M.close(); A.close()
print("ok")
