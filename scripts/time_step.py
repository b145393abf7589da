"""Whole pressure-correction time step on the device (ISPH_THETA = 0: the Helmholtz system is the identity; 0.5 = the
setting the reference's rev390 table was produced with: three right-hand sides through FGMRES + block ILU(0)), 3-D TGV,
ncell^3 particles, device-resident arrays: computePre -> Helmholtz RHS -> Poisson assemble + GMRES/ILU(0) -> zero mean ->
velocity/pressure correction -> advance.  Prints the stage times of a few consecutive steps (ms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import isph_amd
from isph_amd import hip, workload, dist

nc = int(os.environ.get("ISPH_NCELL", "100"))
theta = float(os.environ.get("ISPH_THETA", "0"))
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
nu = torch.from_numpy(parts["nu"]).to(dev)
v = torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)
p = torch.zeros(nall, dtype=torch.float64, device=dev)
zeros3 = torch.zeros((nall, 3), dtype=torch.float64, device=dev)
g = np.zeros(3)
dt = spec.dt


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for step in range(4):
    t0 = sync()
    vf = hip.compute_volumes(ctx, dp, colmap)
    vfrac = vf[own].contiguous()
    t1 = sync()
    hits = 0
    if theta == 0.0:
        # Helmholtz with theta = 0: b is v* (viscous term, body force, -dt/rho grad p)
        H, bh = hip.assemble_helmholtz(ctx, dp, colmap, dt, 0.0, nu, rho, p, zeros3, g, v, vfrac=vfrac, rhs_only=True)
        vstar = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        vstar[:, 0], vstar[:, 1], vstar[:, 2] = bh[:n], bh[n:2 * n], bh[2 * n:3 * n]
    else:
        H, bh = hip.assemble_helmholtz(ctx, dp, colmap, dt, theta, nu, rho, p, zeros3, g, v, vfrac=vfrac)
        MH = hip.Precond(ctx, H, "bjacobi-ilu0", 512)
        xh = torch.empty(3 * n, dtype=torch.float64, device=dev)
        for k in range(3):
            xh[k * n:(k + 1) * n] = v[:n, k]             # initial guess: the current velocity
        ih = hip.solve(ctx, H, bh, xh, prec=MH, singular=False, nvec=3, lda=n)
        hits = ih.iters
        MH.close(); H.close()
        vstar = torch.stack([xh[:n], xh[n:2 * n], xh[2 * n:3 * n]], dim=1).contiguous()
    vstar_all = vstar[own].contiguous()
    t2 = sync()
    A, b = hip.assemble_poisson(ctx, dp, colmap, dt, rho, vstar_all, vfrac=vfrac, ncol=plan.ncol)
    t3 = sync()
    M = hip.Precond(ctx, A, "bjacobi-ilu0", 512)
    dpv = torch.zeros(n, dtype=torch.float64, device=dev)
    info = hip.solve(ctx, A, b, dpv, prec=M, singular=True)
    M.close(); A.close()
    dpv -= dpv.mean()
    t4 = sync()
    dp_all = dpv[own].contiguous()
    hip.correct_velocity_pressure(ctx, dp, colmap, dt, rho, dp_all, vstar_all, p, vfrac)
    dpa = hip.advance_begin(ctx, dp, colmap, dt, p, v, vstar_all, vfrac)
    t5 = sync()
    if os.environ.get("ISPH_SHIFT"):              # fix isph/shift 0.05 (taylor-green-vortex-3d.lmp): timed on copies
        xs, vs_, ps = dp["x"].clone(), v.clone(), p.clone()
        t6 = sync()
        hip.shift_particles(ctx, dp, colmap, 0.05, spec.cut, 0.1, dt, xs, vs_, ps, vfrac)
        print("        particle shift %.1f ms" % ((sync() - t6) * 1e3))
    print("step %d: computePre %.1f  helmholtz %.1f [%d its]  poisson-assemble %.1f  solve(+ILU) %.1f [%d its]  correct+advance %.1f  total %.1f ms"
          % (step, (t1 - t0) * 1e3, (t2 - t1) * 1e3, hits, (t3 - t2) * 1e3, (t4 - t3) * 1e3, info.iters, (t5 - t4) * 1e3, (t5 - t0) * 1e3))
