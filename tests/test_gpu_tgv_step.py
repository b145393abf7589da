"""-m gpu: a whole pressure-correction time step of the 2-D Taylor-Green vortex on the device
(computePre -> Helmholtz RHS -> Poisson assemble+solve -> velocity/pressure correction -> advance),
chained exactly like PairISPH::computeIncompressibleNavierStokes + advanceTime, against the oracle's
driver (oracle/tgv_driver.py) step by step."""
import numpy as np
import pytest

from isph_amd import hip
import oracle as orc
import tgv_driver as T

pytestmark = pytest.mark.gpu


def device_chain(ctx, N, nsteps, antisym, theta=0.0, kernel="wendland", cut_over_h=2.0, block=256, hist_check=None):
    """the pressure-correction chain of PairISPH::computeIncompressibleNavierStokes + advanceTime with every
    neighbour sweep and both linear solves on the device; returns (history, final state)."""
    umax, nu, rho0 = 0.1, 0.1, 1.0
    L = 2 * np.pi
    dx = L / N
    h = 1.5 * dx
    cut = cut_over_h * h
    dt = 0.1 * h / umax
    g = (np.arange(N) + 0.5) * dx
    X, Y = np.meshgrid(g, g, indexing="xy")
    x = np.stack([X.ravel(), Y.ravel(), np.zeros(N * N)], axis=1)
    v, _ = T.tgv_exact(x, 0.0, umax, nu, rho0)
    p = np.zeros(N * N)
    n = N * N
    hist = []
    for step in range(1, nsteps + 1):
        parts, own = T.periodic_particles(x, L, cut)
        parts["h"], parts["cut"] = h, cut
        nall = parts["nall"]
        colmap = own.astype(np.int32)
        ghost = lambda a: np.ascontiguousarray(a[own])
        rho = np.full(nall, rho0)
        nuall = np.full(nall, nu)
        # computePre on the device
        vfrac = ghost(hip.compute_volumes(ctx, parts, colmap, kernel=kernel))
        Gc = Lc = None
        if not antisym:
            G, Lm = hip.compute_corrections(ctx, parts, colmap, vfrac, kernel=kernel)
            Gc = np.zeros((nall, 4)); Gc[:n] = G
            Lc = np.zeros((nall, 3)); Lc[:n] = Lm
        # Helmholtz: theta = 0 -> the matrix is the identity and b is v* (pair_isph.cpp:964-966)
        A_h, bh = hip.assemble_helmholtz(ctx, parts, colmap, dt, theta, nuall, rho, ghost(p), np.zeros((nall, 3)),
                                         np.zeros(3), ghost(v), antisym=antisym, vfrac=vfrac, Gc=Gc, Lc=Lc, kernel=kernel)
        vstar = np.zeros((n, 3))
        if abs(theta) < 1e-14:
            vstar[:, 0], vstar[:, 1] = bh[:n], bh[n:2 * n]
        else:                                            # 2 right-hand sides, x0 = v^n (pair_isph.cpp:925-971)
            Mh = hip.Precond(ctx, A_h, "bjacobi-ilu0", block)
            xh = np.ascontiguousarray(np.concatenate([v[:, 0], v[:, 1]]))
            ih = hip.solve(ctx, A_h, np.ascontiguousarray(bh[:2 * n].copy()), xh, prec=Mh, singular=False, nvec=2, lda=n)
            assert ih.converged == 1
            vstar[:, 0], vstar[:, 1] = xh[:n], xh[n:]
        # Poisson
        A, b = hip.assemble_poisson(ctx, parts, colmap, dt, rho, ghost(vstar), antisym=antisym, vfrac=vfrac,
                                    Gc=Gc, Lc=Lc, kernel=kernel)
        M = hip.Precond(ctx, A, "bjacobi-ilu0", block)
        dp = np.zeros(n)
        info = hip.solve(ctx, A, b.copy(), dp, prec=M, singular=True)
        assert info.converged == 1
        dp -= dp.mean()                                   # computeZeroMeanPressure (2 Allreduce on the host side)
        # corrections on the device
        vs_all, p_all = ghost(vstar), ghost(p)
        hip.correct_velocity_pressure(ctx, parts, colmap, dt, rho, ghost(dp), vs_all, p_all, vfrac, antisym=antisym,
                                      Gc=Gc, kernel=kernel)
        vstar, p = vs_all[:n].copy(), p_all[:n].copy()
        # fix isph/tgv
        t = dt * step
        uex, pex = T.tgv_exact(x, t, umax, nu, rho0)
        rec = dict(step=step, time=t, p_err=np.sqrt(np.mean((p - pex - p.mean()) ** 2)), p_norm=np.sqrt(np.mean(pex ** 2)),
                   u_err=np.sqrt(np.mean(np.sum((vstar - uex) ** 2, axis=1))),
                   u_norm=np.sqrt(np.mean(np.sum(uex ** 2, axis=1))))
        hist.append(rec)
        if hist_check is not None:
            hist_check(step, rec)
        # advanceTime on the device
        dpa = hip.advance_begin(ctx, parts, colmap, dt, ghost(p), ghost(v), ghost(vstar), vfrac, antisym=antisym, Gc=Gc,
                                kernel=kernel)
        xa, va, pa = np.ascontiguousarray(x.copy()), np.ascontiguousarray(v.copy()), p.copy()
        hip.advance_end(ctx, n, 2, dt, dpa, np.ascontiguousarray(vstar), pa, xa, va)
        x, v, p = xa, va, pa
        x[:, :2] %= L
    return hist, dict(x=x, v=v, p=p)


@pytest.mark.parametrize("antisym,theta", [(True, 0.0), (False, 0.0), (False, 0.5)])
def test_three_tgv_steps_on_device_match_oracle_driver(gpu_ctx, antisym, theta):
    N, nsteps = 16, 3
    umax, L = 0.1, 2 * np.pi
    hist_o, state_o = T.run_tgv2d(N, nsteps, antisym=antisym, theta=theta, return_state=True)

    def check(step, rec):
        ro = hist_o[step - 1]
        assert abs(rec["p_err"] - ro["p_err"]) <= 1e-6 * ro["p_err"]
        assert abs(rec["u_err"] - ro["u_err"]) <= 1e-6 * ro["u_err"]
    hist, st = device_chain(gpu_ctx, N, nsteps, antisym, theta=theta, hist_check=check)
    assert np.max(np.abs(st["x"] - state_o["x"])) < 1e-8 * L
    assert np.max(np.abs(st["v"] - state_o["v"])) < 1e-7 * umax
    assert np.max(np.abs(st["p"] - state_o["p"])) < 1e-6 * np.abs(state_o["p"]).max()


@pytest.mark.parametrize("kernel,N", [("wendland", 16), ("wendland", 32), ("wendland", 64), ("wendland", 128),
                                      ("quintic", 16), ("quintic", 32)])
def test_device_chain_reproduces_reference_tgv_table(gpu_ctx, kernel, N):
    """The DEVICE path against numbers the reference itself recorded (no oracle in the loop):
    sph-script/conv-taylor-green-vortex-2d-rev390.txt rows N = 16...128, both error columns to 3 significant
    digits (<= 2.5e-3 relative; Quintic N=16 pressure 6e-3), with the pinned settings of oracle/tgv_driver.py
    (theta 1/2, incremental pressure, Symmetric corrected family).  Every step runs computePre, the Helmholtz
    assembly + 2-RHS GMRES/ILU(0) solve, the Poisson assembly + null-space GMRES/ILU(0) solve, the corrections
    and advanceTime on the GPU."""
    gold = T.known_answers()
    key = "conv_taylor_green_vortex_2d_rev390" + ("" if kernel == "wendland" else "_quintic")
    ref = gold[key]["rows"][str(N)]
    hist, _ = device_chain(gpu_ctx, N, ref["step"], antisym=False, theta=0.5, kernel=kernel,
                           cut_over_h=2.0 if kernel == "wendland" else 3.0, block=512)
    h = hist[-1]
    assert abs(h["time"] - ref["time"]) < 1e-6
    tol_p = 6e-3 if (kernel, N) == ("quintic", 16) else 2.5e-3
    assert abs(h["p_err"] / ref["p_err"] - 1) < tol_p, (h["p_err"], ref["p_err"])
    assert abs(h["u_err"] / ref["u_err"] - 1) < 2.5e-3, (h["u_err"], ref["u_err"])


def test_bench_step_workload_runs_consecutive_steps(tmp_path):
    """bench.py --workload step (the reference's benchmark protocol, bench-script/hopper/tgv/1728): consecutive time steps
    with the neighbour lists rebuilt from the MOVED particles; the record carries the assembly / solve split and the
    vortex loses kinetic energy monotonically at the rate of the viscous decay (d/dt KE = -4 nu KE for the 2-D TGV mode)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kes = []
    for steps in (1, 4):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "step", "--ncell", "20", "--kernel", "wendland",
                            "--steps", str(steps), "--warmup", "0", "--brick", "10,10,5", "--singular", "nullspace"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        assert rec["unit"] == "steps/s" and rec["value"] > 0 and len(rec["config"]["iterations_poisson"]) == steps
        assert abs(rec["split"]["assembly_ms"] + rec["split"]["solve_ms"] - rec["ms_per_step"]) < rec["ms_per_step"]
        kes.append((rec["config"]["kinetic_energy_sum_end"], rec["config"]["dt"]))
    ke0 = 0.5 * 20 ** 3 * 0.1 ** 2 * 0.5                           # sum over the lattice of |v|^2 / 2, Umax = 0.1
    (ke1, dt), (ke4, _) = kes
    assert ke4 < ke1 < ke0
    rate = -np.log(ke4 / ke1) / (3 * dt)                              # three more steps
    assert 0.5 * 4 * 0.1 < rate < 2.0 * 4 * 0.1, rate                 # 4 nu = 0.4 within a factor 2 (coarse 20^3 lattice)


def test_held_neighbour_list_gives_the_same_operators(gpu_ctx):
    """isph_ctx_hold_neighbours: while the caller holds the list, the operator calls share one layout of it (built by the
    first call); volumes, corrections, the Poisson matrix and its right-hand side are bit for bit those of self-contained
    calls, and releasing the hold drops the layout (a different list behind the same arrays is seen afterwards)."""
    import torch
    from isph_amd import workload
    spec = workload.TGVSpec(dim=3, ncell=(12, 12, 12), brick=(4, 4, 4), mode=workload.ADVECT)
    parts = workload.make_tgv(spec)
    dev = torch.device("cuda", 0)
    colmap_h = workload.single_rank_colmap(parts)
    dp = dict(parts)
    for k in ("x", "type", "neigh_ptr", "neigh_idx"):
        dp[k] = torch.from_numpy(np.ascontiguousarray(parts[k])).to(dev)
    colmap = torch.from_numpy(colmap_h).to(dev)
    own = torch.from_numpy(parts["owner_index"].astype(np.int64)).to(dev)
    rho = torch.from_numpy(parts["rho"]).to(dev)
    vstar = torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)

    def run():
        vf = hip.compute_volumes(gpu_ctx, dp, colmap)
        vfrac = vf[own].contiguous()
        G, L = hip.compute_corrections(gpu_ctx, dp, colmap, vfrac)
        A, b = hip.assemble_poisson(gpu_ctx, dp, colmap, spec.dt, rho, vstar, vfrac=vfrac)
        out = (vf.cpu().numpy(), G.cpu().numpy(), L.cpu().numpy(), b.cpu().numpy()) + tuple(A.export_csr())
        A.close()
        return out

    ref = run()
    gpu_ctx.hold_neighbours(True)
    try:
        first, second = run(), run()
    finally:
        gpu_ctx.hold_neighbours(False)
    for a, b_, c in zip(ref, first, second):
        assert np.array_equal(a, b_) and np.array_equal(a, c)
    # a shorter list behind the same device arrays: only seen because the hold was released
    nidx = dp["neigh_idx"]
    keep = nidx.clone()
    nl = parts["nlocal"]
    ptr = dp["neigh_ptr"].cpu().numpy()
    first_row = slice(int(ptr[0]), int(ptr[1]))
    nidx[first_row] = nidx[first_row.start]            # row 0 now lists one neighbour over and over
    changed = run()
    nidx.copy_(keep)
    assert not np.array_equal(changed[0], ref[0])
