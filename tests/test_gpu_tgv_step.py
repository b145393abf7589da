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


@pytest.mark.parametrize("antisym", [True, False])
def test_three_tgv_steps_on_device_match_oracle_driver(gpu_ctx, antisym):
    N, nsteps = 16, 3
    umax, nu, rho0 = 0.1, 0.1, 1.0
    L = 2 * np.pi
    dx = L / N
    h = 1.5 * dx
    cut = 2.0 * h
    dt = 0.1 * h / umax
    hist_o, state_o = T.run_tgv2d(N, nsteps, antisym=antisym, return_state=True)

    g = (np.arange(N) + 0.5) * dx
    X, Y = np.meshgrid(g, g, indexing="xy")
    x = np.stack([X.ravel(), Y.ravel(), np.zeros(N * N)], axis=1)
    v, _ = T.tgv_exact(x, 0.0, umax, nu, rho0)
    p = np.zeros(N * N)
    n = N * N
    for step in range(1, nsteps + 1):
        parts, own = T.periodic_particles(x, L, cut)
        parts["h"], parts["cut"] = h, cut
        nall = parts["nall"]
        colmap = own.astype(np.int32)
        ghost = lambda a: np.ascontiguousarray(a[own])
        rho = np.full(nall, rho0)
        nuall = np.full(nall, nu)
        # computePre on the device
        vfrac = ghost(hip.compute_volumes(gpu_ctx, parts, colmap))
        Gc = Lc = None
        if not antisym:
            G, Lm = hip.compute_corrections(gpu_ctx, parts, colmap, vfrac)
            Gc = np.zeros((nall, 4)); Gc[:n] = G
            Lc = np.zeros((nall, 3)); Lc[:n] = Lm
        # Helmholtz with theta = 0: the matrix is the identity and b is v*
        A_h, bh = hip.assemble_helmholtz(gpu_ctx, parts, colmap, dt, 0.0, nuall, rho, ghost(p), np.zeros((nall, 3)),
                                         np.zeros(3), ghost(v), antisym=antisym, vfrac=vfrac, Gc=Gc, Lc=Lc)
        vstar = np.zeros((n, 3))
        vstar[:, 0], vstar[:, 1] = bh[:n], bh[n:2 * n]
        # Poisson
        A, b = hip.assemble_poisson(gpu_ctx, parts, colmap, dt, rho, ghost(vstar), antisym=antisym, vfrac=vfrac,
                                    Gc=Gc, Lc=Lc)
        M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 256)
        dp = np.zeros(n)
        info = hip.solve(gpu_ctx, A, b.copy(), dp, prec=M, singular=True)
        assert info.converged == 1
        dp -= dp.mean()                                   # computeZeroMeanPressure (2 Allreduce on the host side)
        # corrections on the device
        vs_all, p_all = ghost(vstar), ghost(p)
        hip.correct_velocity_pressure(gpu_ctx, parts, colmap, dt, rho, ghost(dp), vs_all, p_all, vfrac, antisym=antisym,
                                      Gc=Gc)
        vstar, p = vs_all[:n].copy(), p_all[:n].copy()
        # fix isph/tgv
        t = dt * step
        uex, pex = T.tgv_exact(x, t, umax, nu, rho0)
        p_err = np.sqrt(np.mean((p - pex - p.mean()) ** 2))
        u_err = np.sqrt(np.mean(np.sum((vstar - uex) ** 2, axis=1)))
        ro = hist_o[step - 1]
        assert abs(p_err - ro["p_err"]) <= 1e-6 * ro["p_err"]
        assert abs(u_err - ro["u_err"]) <= 1e-6 * ro["u_err"]
        # advanceTime on the device
        dpa = hip.advance_begin(gpu_ctx, parts, colmap, dt, ghost(p), ghost(v), ghost(vstar), vfrac, antisym=antisym, Gc=Gc)
        xa, va, pa = np.ascontiguousarray(x.copy()), np.ascontiguousarray(v.copy()), p.copy()
        hip.advance_end(gpu_ctx, n, 2, dt, dpa, np.ascontiguousarray(vstar), pa, xa, va)
        x, v, p = xa, va, pa
        x[:, :2] %= L
    assert np.max(np.abs(x - state_o["x"])) < 1e-8 * L
    assert np.max(np.abs(v - state_o["v"])) < 1e-7 * umax
    assert np.max(np.abs(p - state_o["p"])) < 1e-6 * np.abs(state_o["p"]).max()
