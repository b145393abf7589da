"""2-D Taylor-Green-vortex mini driver on top of the CPU ORACLE (test infrastructure).

Purpose: pin the oracle's assembly + solve restatement against the reference's
own recorded known answers, sph-script/conv-taylor-green-vortex-2d-rev390.txt
("TAYLOR-GREEN-VORTEX-2D.LMP, SINGULAR POISSON NULLSPACE, NO SHIFT", Wendland):

    N = 16  step 3  t=1.767146  pressure l2 error 8.466849370245e-04 (1.23340621e-03)
                                velocity l2 error 7.500246669496e-04 (4.96568611e-02)
    N = 32  step 6  t=1.767146  pressure 1.995025956346e-04, velocity 1.695211327348e-04

It restates, per time step, the pressure-correction scheme of
PairISPH::computeIncompressibleNavierStokes (pair_isph.cpp:910-1034) with
theta = 0 (taylor-green-vortex.xml:15):
  computePre (volumes [+ G_i, L_i])             pair_isph_corrected.cpp:302-369
  Helmholtz RHS, theta=0 => v* = b              functor_incomp_navier_stokes_helmholtz.h:52-159
  Poisson  A dp = -div v*  (NullSpace)          functor_incomp_navier_stokes_poisson.h:52-181
  zero-mean dp                                  pair_isph.cpp:422-464
  v* -= dt/rho grad dp ; p += dp                functor_correct_velocity.h, functor_correct_pressure.h
  error vs analytic TGV (fix isph/tgv runs before fix isph in final_integrate)   fix_isph_tgv.cpp:43-125
  advanceTime: p += grad p . dx ; x += dx ; v = v*   functor_advance_time_{begin,end}.h
The time step of that revision is dt = 0.1 h / Umax.
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import oracle as orc  # noqa: E402

TABLE_REV390_WENDLAND = {
    16: dict(step=3, time=1.767146, p_err=8.466849370245e-04, p_norm=1.23340621e-03,
             u_err=7.500246669496e-04, u_norm=4.96568611e-02),
    32: dict(step=6, time=1.767146, p_err=1.995025956346e-04, p_norm=1.23259792e-03,
             u_err=1.695211327348e-04, u_norm=4.96687154e-02),
    64: dict(step=13, time=1.914408, p_err=7.140008948534e-05, p_norm=1.16213433e-03,
             u_err=3.622266617824e-05, u_norm=4.82266618e-02),
}


def periodic_particles(x, L, cut):
    """LAMMPS-like atom arrays for a fully periodic 2-D box: owned atoms + the
    ghost images within `cut` of the box, full neighbour list by brute force."""
    n = len(x)
    shifts = [(sx, sy) for sx in (-1, 0, 1) for sy in (-1, 0, 1) if (sx, sy) != (0, 0)]
    xs, owner = [x], [np.arange(n)]
    for sx, sy in shifts:
        xi = x + np.array([sx * L, sy * L, 0.0])
        keep = (xi[:, 0] > -cut) & (xi[:, 0] < L + cut) & (xi[:, 1] > -cut) & (xi[:, 1] < L + cut)
        xs.append(xi[keep])
        owner.append(np.nonzero(keep)[0])
    xall = np.concatenate(xs)
    own = np.concatenate(owner)
    nall = len(xall)
    ptr = [0]
    idx = []
    cutsq = cut * cut
    for i in range(n):
        d = xall[:, :2] - xall[i, :2]
        r2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        nb = np.nonzero(r2 < cutsq)[0]
        nb = nb[nb != i]
        idx.append(nb)
        ptr.append(ptr[-1] + len(nb))

    class _Spec:
        rank = 0
    return dict(spec=_Spec(), dim=2, nlocal=n, nall=nall, x=np.ascontiguousarray(xall),
                type=np.ones(nall, np.int32), tag=(own + 1).astype(np.int32),
                owner_rank=np.zeros(nall, np.int32), owner_index=own.astype(np.int32),
                neigh_ptr=np.asarray(ptr, np.int32), neigh_idx=np.concatenate(idx).astype(np.int32)), own


def tgv_exact(x, t, umax, nu, rho):
    u = np.zeros((len(x), 3))
    u[:, 0] = umax * np.exp(-2 * nu * t) * np.sin(x[:, 0]) * np.cos(x[:, 1])
    u[:, 1] = -umax * np.exp(-2 * nu * t) * np.cos(x[:, 0]) * np.sin(x[:, 1])
    p = 0.25 * rho * umax ** 2 * np.exp(-4 * nu * t) * (np.cos(2 * x[:, 0]) + np.cos(2 * x[:, 1]))
    return u, p


def run_tgv2d(N, nsteps, antisym=True, umax=0.1, nu=0.1, rho0=1.0, kernel="wendland", cut_over_h=2.0,
              dt=None, prec="ilu", verbose=False, return_state=False):
    L = 2 * np.pi
    dx = L / N
    h = 1.5 * dx
    cut = cut_over_h * h
    dt = 0.1 * h / umax if dt is None else dt
    g = (np.arange(N) + 0.5) * dx
    X, Y = np.meshgrid(g, g, indexing="xy")
    x = np.stack([X.ravel(), Y.ravel(), np.zeros(N * N)], axis=1)
    v, _ = tgv_exact(x, 0.0, umax, nu, rho0)
    p = np.zeros(N * N)
    n = N * N
    hist = []
    for step in range(1, nsteps + 1):
        parts, own = periodic_particles(x, L, cut)
        parts["h"], parts["cut"] = h, cut
        P = orc.Particles(parts, own, kernel=kernel)
        P.precompute(corrections=not antisym)
        if not antisym:                           # ghosts carry the owner's tensors only through row i: none needed
            pass
        nall = parts["nall"]
        rho = np.full(nall, rho0)
        ghost = lambda a: np.ascontiguousarray(a[own])     # forward_comm_pair
        vall, pall = ghost(v), ghost(p)
        # ---- Helmholtz RHS with theta = 0:  v* = v + (dt nu lap v) - dt/rho grad p
        w = P.laplacian_apply(vall, antisym, dt, material=np.full(nall, nu * rho0), filt=(orc.FLUID, orc.ALL)) / rho0
        gp = P.gradient(pall, antisym, filt=(orc.FLUID, orc.FLUID))
        vstar = v + w - dt / rho0 * gp
        vstar[:, 2] = 0.0
        # ---- Poisson
        rp, ci, val, b = P.poisson(dt, rho, ghost(vstar), antisym=antisym, singular=orc.NULLSPACE)
        ilu = orc.ILU(rp, ci, val, 0) if prec == "ilu" else None
        dp, info, _ = orc.solve(rp, ci, val, b, singular=True, prec=prec, ilu=ilu)
        assert info.converged, "Poisson solve did not converge"
        dp -= dp.mean()                           # computeZeroMeanPressure
        # ---- corrections
        gdp = P.gradient(ghost(dp), antisym, filt=(orc.FLUID, orc.FLUID))
        vstar = vstar - dt / rho0 * gdp
        vstar[:, 2] = 0.0
        p = p + dp
        # ---- fix isph/tgv: error at t = dt*step, positions not yet advanced
        t = dt * step
        uex, pex = tgv_exact(x, t, umax, nu, rho0)
        pavg = p.mean()
        rec = dict(step=step, time=t, iters=info.iters,
                   p_err=np.sqrt(np.mean((p - pex - pavg) ** 2)), p_norm=np.sqrt(np.mean(pex ** 2)),
                   u_err=np.sqrt(np.mean(np.sum((vstar - uex) ** 2, axis=1))), u_norm=np.sqrt(np.mean(np.sum(uex ** 2, axis=1))))
        hist.append(rec)
        if verbose:
            print(rec)
        # ---- advanceTime
        dxp = 0.5 * dt * (vstar + v)
        gpn = P.gradient(ghost(p), antisym, filt=(orc.FLUID, orc.FLUID))
        p = p + np.sum(gpn * dxp, axis=1)
        x = x + dxp
        x[:, :2] %= L
        v = vstar
    if return_state:
        return hist, dict(x=x, v=v, p=p)
    return hist


if __name__ == "__main__":
    for N in (16, 32):
        ref = TABLE_REV390_WENDLAND[N]
        for antisym in (True, False):
            h = run_tgv2d(N, ref["step"], antisym=antisym)[-1]
            print("N=%d %s  t=%.6f  p_err %.6e (ref %.6e)  u_err %.6e (ref %.6e)  norms %.6e %.6e" %
                  (N, "AntiSym" if antisym else "Sym", h["time"], h["p_err"], ref["p_err"], h["u_err"], ref["u_err"],
                   h["p_norm"], h["u_norm"]))
