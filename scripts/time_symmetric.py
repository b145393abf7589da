"""Assembly cost of the Symmetric (corrected, G_i / L_i) operator family at 100^3 beside the AntiSymmetric default.
usage on the GPU box: python scripts/time_symmetric.py [ncell]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import isph_amd
from isph_amd import hip, workload, dist

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = hip.Context(0, stream=st.cuda_stream)
spec = workload.TGVSpec(dim=3, ncell=(nc, nc, nc), brick=(8, 8, 8), mode=workload.ADVECT)
parts = workload.make_tgv(spec)
plan = dist.make_plan(parts, None)
n, nall = parts["nlocal"], parts["nall"]
dp = dict(parts)
for k in ("x", "type", "neigh_ptr", "neigh_idx"):
    dp[k] = torch.from_numpy(np.ascontiguousarray(parts[k])).to(dev)
colmap = torch.from_numpy(plan.colmap).to(dev)
own = torch.from_numpy(parts["owner_index"].astype(np.int64)).to(dev)
rho = torch.from_numpy(parts["rho"]).to(dev)
v = torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    t0 = sync()
    vf = hip.compute_volumes(ctx, dp, colmap)
    vfrac = vf[own].contiguous()
    t1 = sync()
    Gc, Lc = hip.compute_corrections(ctx, dp, colmap, vfrac)
    Gca, Lca = Gc[own].contiguous(), Lc[own].contiguous()
    t2 = sync()
    A, b = hip.assemble_poisson(ctx, dp, colmap, spec.dt, rho, v, antisym=False, vfrac=vfrac, Gc=Gca, Lc=Lca)
    t3 = sync()
    A2, b2 = hip.assemble_poisson(ctx, dp, colmap, spec.dt, rho, v, antisym=True, vfrac=vfrac)
    t4 = sync()
    M = hip.Precond(ctx, A, "bjacobi-ilu0", 512)
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    info = hip.solve(ctx, A, b.clone(), x, prec=M, singular=True)
    t5 = sync()
    print("rep %d: volumes %.2f  corrections (G_i, L_i) %.2f  Poisson assembly Symmetric %.2f | AntiSymmetric %.2f  "
          "Symmetric solve(+ILU) %.1f ms [%d its, converged %d]" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3,
                                                                     (t5 - t4) * 1e3, info.iters, info.converged), flush=True)
    M.close(); A.close(); A2.close()
