"""BASELINE configs[3]-shaped step (closed lid-driven cavity, 114^3 fluid + 6 wall layers = 2.0 M particles): stage times
of computePre, the 3x3 block Helmholtz system (Navier-slip walls) and the pressure Poisson system with wall Neumann rows.
usage: python scripts/run_config3.py [fluid cells per side] [bjacobi-ilu0 | sa-amg] [brick, e.g. 9,9,6]
A brick that tiles the lattice (fluid + 2 x 6 wall cells per side) and holds <= 1024 particles makes the bricks the
block-Jacobi subdomains (isph_prec_create_blocks) instead of 512 consecutive rows of the 8x8x8 numbering.
(lid-driven-cavity.xml:30-32 selects ML: sa-amg is the reference's setting for this case)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import isph_amd  # noqa: F401
from isph_amd import hip, workload

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 114
prec = sys.argv[2] if len(sys.argv) > 2 else "bjacobi-ilu0"
THETA, BETA = 0.5, 0.0
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = hip.Context(0, stream=st.cuda_stream)
brick = tuple(int(t) for t in sys.argv[3].split(",")) if len(sys.argv) > 3 else (8, 8, 8)
p = workload.make_cavity(nf, wall=6, brick=brick)
n, nall = p["nlocal"], p["nall"]
brows = brick[0] * brick[1] * brick[2]
bptr = np.arange(0, n + brows, brows).clip(0, n).astype(np.int32) if all((nf + 12) % k == 0 for k in brick) and brows <= 1024 else None
colmap_h = workload.single_rank_colmap(p)
dp = dict(p)
for k in ("x", "type", "neigh_ptr", "neigh_idx"):
    dp[k] = torch.from_numpy(np.ascontiguousarray(p[k])).to(dev)
colmap = torch.from_numpy(colmap_h).to(dev)
own = torch.from_numpy(p["owner_index"].astype(np.int64)).to(dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rho, nu, vel, nrm = t(p["rho"]), t(p["nu"]), t(p["v"]), t(p["normal"])
pres = torch.zeros(nall, dtype=torch.float64, device=dev)
force = torch.zeros((nall, 3), dtype=torch.float64, device=dev)
mask = (p["type"][:n] == 1).astype(np.int32)


def sync():
    torch.cuda.synchronize(); ctx.sync()
    return time.perf_counter()


print("particles %d, preconditioner %s, subdomains %s" % (n, prec, ("%dx%dx%d bricks" % brick) if bptr is not None else "512 consecutive rows"), flush=True)
for step in range(3):
    t0 = sync()
    vf = hip.compute_volumes(ctx, dp, colmap)
    vfrac = vf[own].contiguous()
    G, _ = hip.compute_corrections(ctx, dp, colmap, vfrac)
    Gc = G[own].contiguous()
    t1 = sync()
    blocks, b = hip.assemble_block_helmholtz(ctx, dp, colmap, p["dt"], THETA, BETA, nu, rho, pres, force, np.zeros(3),
                                             vel, normal=nrm, vfrac=vfrac, Gc=Gc, kinds=p["kinds"])
    t2 = sync()
    x = vel[:n].t().contiguous().reshape(-1).clone()
    M = hip.PrecondAMG(ctx, blocks[0][0], params=hip.AmgParams(block=512)) if prec == "sa-amg" else (hip.Precond(ctx, blocks[0][0], prec, block_ptr=bptr) if bptr is not None and prec == "bjacobi-ilu0" else hip.Precond(ctx, blocks[0][0], prec, 512))
    info = hip.solve_block(ctx, blocks, b.clone(), x, prec=M)
    t3 = sync()
    vs = x.reshape(3, n).t().contiguous()
    vstar = torch.zeros((nall, 3), dtype=torch.float64, device=dev)
    vstar[:n] = vs
    vstar = vstar[own].contiguous()
    A, bp = hip.assemble_poisson(ctx, dp, colmap, p["dt"], rho, vstar, vfrac=vfrac, Gc=Gc, kinds=p["kinds"], normal=nrm)
    if step == 0:
        im = A.info()
        print("Poisson operator: %d rows, %d entries (%.2f per row), SpMV algorithmic bytes %.4f GB" %
              (im["nrow"], im["nnz"], im["nnz"] / im["nrow"], (12 * im["nnz"] + 16 * im["nrow"] + 4 * (im["nrow"] + 1)) / 1e9), flush=True)
    t4 = sync()
    xp = torch.zeros(n, dtype=torch.float64, device=dev)
    if prec == "sa-amg":
        nvh = mask.astype(np.float64) / np.sqrt(float(mask.sum()))
        MP = hip.PrecondAMG(ctx, A, nullvec=torch.from_numpy(nvh).to(dev), params=hip.AmgParams(block=512))
    else:
        MP = hip.Precond(ctx, A, prec, block_ptr=bptr) if bptr is not None and prec == "bjacobi-ilu0" else hip.Precond(ctx, A, prec, 512)
    ip = hip.solve(ctx, A, bp.clone(), xp, prec=MP, singular=True, null_mask=mask)
    t5 = sync()
    print("step %d: computePre %.1f  block-Helmholtz assembly %.1f  block solve(+ILU) %.1f [%d its, conv %d]  Poisson assembly %.1f  "
          "Poisson solve(+ILU) %.1f [%d its, conv %d]  total %.1f ms" % (step, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, info.iters,
          info.converged, (t4 - t3) * 1e3, (t5 - t4) * 1e3, ip.iters, ip.converged, (t5 - t0) * 1e3), flush=True)
    M.close(); MP.close(); A.close()
    for row in blocks:
        for B in row:
            if B is not None:
                B.close()
