"""BASELINE configs[0]: 2-D Taylor-Green lattice, 128^2 = 16384 particles, Wendland, CG + ILU(0) (the reference's own
CPU-runnable case): GPU solve time (preconditioner build + Block CG, tol 1e-6 as USER-REAXC-T/solver_lin_belos.h:236-245)
beside the oracle on the host cores.  Small-problem regime: the GPU is launch-latency bound here."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import isph_amd
from isph_amd import hip, workload
import oracle as orc
from problems import Problem, tgv_spec

n = int(os.environ.get("ISPH_NCELL", "128"))
pr = Problem(tgv_spec(dim=2, n=n, mode=workload.LATTICE))
rp, ci, val, _ = pr.poisson()
x = pr.parts["x"][:pr.n]
b = np.cos(2 * x[:, 0]) + np.cos(2 * x[:, 1])
bs = 512
bp = np.arange(0, pr.n + bs, bs).clip(0, pr.n).astype(np.int32)
prm_o = orc.SolverParams(solver_type=1, tol=1e-6)
t0 = time.perf_counter()
ilu = orc.ILU(rp, ci, val, 0, bp)
xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="ilu", ilu=ilu, params=prm_o)
t_cpu = time.perf_counter() - t0
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
A = hip.Matrix.from_csr(ctx, rp, ci, val)
dev = torch.device("cuda", 0)
bd = torch.from_numpy(b).to(dev)
prm = hip.SolverParams(solver_type=1, tol=1e-6)
ts = []
for rep in range(6):
    bw, xg = bd.clone(), torch.zeros(pr.n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    M = hip.Precond(ctx, A, "bjacobi-ilu0", bs)
    info = hip.solve(ctx, A, bw, xg, prec=M, singular=True, params=prm)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    M.close()
err = np.linalg.norm(xg.cpu().numpy() - xo) / np.linalg.norm(xo)
print("rows %d nnz %d | CPU oracle (%d threads) %.1f ms, %d its | GPU %.2f ms (min of 5), %d its | rel diff %.2e"
      % (pr.n, len(val), orc.num_threads(), t_cpu * 1e3, io.iters, min(ts[1:]) * 1e3, info.iters, err))
