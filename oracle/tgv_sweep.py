"""Sweep of the settings the reference's 2-D TGV table does not record
(sph-script/conv-taylor-green-vortex-2d-rev390.txt:1 only says "NULLSPACE, NO SHIFT").

TEST INFRASTRUCTURE (oracle side).  The pressure-correction step of
PairISPH::computeIncompressibleNavierStokes (pair_isph.cpp:910-1034) is restated
around the CPU oracle with every unrecorded choice as a switch:

  theta        0 | 0.5 | 1       ("theta", pair_isph.cpp:1753; xml today: 0)
  incremental  True | False      ("Use Incremental Pressure", pair_isph.cpp:1776, functor_correct_pressure.h:37,
                                  functor_incomp_navier_stokes_helmholtz.h:139)
  antisym      True | False      ("Use Momentum Preserve Operator", pair_isph.cpp:1779, pair_isph_corrected.cpp:48)
  err_after    False | True      fix isph/tgv before | after fix isph's advanceTime (fix order in the script)
  err_on       "vstar" | "v"     fix_isph_tgv.cpp:53-57
  dtmode       "0.1h" | "0.05dx" taylor-green-vortex-3d.lmp:27 | taylor-green-vortex-2d.lmp:29
  origin       0.5 | 0.0         lattice origin (…-2d.lmp:69 | …-3d.lmp:66)
  advance      "trap" | "new" | "old"   x += dt/2 (v*+v) (functor_advance_time_end.h) | dt v* | dt v
  padvect      True | False      p += grad p . dx (functor_advance_time_begin.h)
  shift        0 | c             fix isph/shift c (fix_isph_shift.cpp), shift_v "max" | "mean" speed

usage: python tgv_sweep.py [kernel]   -> prints the sweep table for N = 16, 32 (kept as oracle/tgv_sweep_rev390.txt)
Result: exactly one cell fits (score = worst relative deviation of the 2 errors + 2 norms over N = 16, 32):
theta 0.5 / incremental / Symmetric / error before advanceTime / trapezoid / pressure advected: 1.8e-3;
the runner-up is at 2e-1.  tgv_driver.PINNED holds it.
"""
import itertools
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import oracle as orc  # noqa: E402
from tgv_driver import periodic_particles, tgv_exact  # noqa: E402

from tgv_driver import known_answers  # noqa: E402

_G = known_answers()
REV390 = {"wendland": {int(k): v for k, v in _G["conv_taylor_green_vortex_2d_rev390"]["rows"].items()},
          "quintic": {int(k): v for k, v in _G["conv_taylor_green_vortex_2d_rev390_quintic"]["rows"].items()}}


def run(N, nsteps, theta=0.0, incremental=True, antisym=True, err_after=False, err_on="vstar", dtmode="0.1h",
        origin=0.5, advance="trap", padvect=True, kernel="wendland", cut_over_h=None, corrections=None,
        umax=0.1, nu=0.1, rho0=1.0, tol=1e-10, all_steps=False, shift=0.0, shiftcut_over_cut=1.0, shift_seq=True, shift_nointerp=False, shift_v='max'):
    L = 2 * np.pi
    dx = L / N
    h = 1.5 * dx
    if cut_over_h is None:
        cut_over_h = 2.0 if kernel == "wendland" else 3.0
    cut = cut_over_h * h
    dt = 0.1 * h / umax if dtmode == "0.1h" else 0.05 * dx / umax
    g = (np.arange(N) + origin) * dx
    X, Y = np.meshgrid(g, g, indexing="xy")
    x = np.stack([X.ravel(), Y.ravel(), np.zeros(N * N)], axis=1)
    v, _ = tgv_exact(x, 0.0, umax, nu, rho0)
    p = np.zeros(N * N)
    if corrections is None:
        corrections = not antisym
    prm = orc.SolverParams(tol=tol)
    out = []
    for step in range(1, nsteps + 1):
        parts, own = periodic_particles(x, L, cut)
        parts["h"], parts["cut"] = h, cut
        P = orc.Particles(parts, own, kernel=kernel)
        P.precompute(corrections=corrections)
        nall = parts["nall"]
        rho = np.full(nall, rho0)
        ghost = lambda a: np.ascontiguousarray(a[own])
        vall, pall = ghost(v), ghost(p)
        mat = np.full(nall, nu * rho0)
        # Helmholtz (functor_incomp_navier_stokes_helmholtz.h:52-159):  (I - theta dt nu lap) v* = v + (1-theta) dt nu lap v - dt/rho grad p
        lap_v = P.laplacian_apply(vall, antisym, dt, material=mat, filt=(orc.FLUID, orc.ALL)) / rho0
        rhs = v + (1.0 - theta) * lap_v
        if incremental:
            rhs = rhs - dt / rho0 * P.gradient(pall, antisym, filt=(orc.FLUID, orc.FLUID))
        if theta < 1e-14:
            vstar = rhs
        else:
            rp, ci = P.graph()
            val = P.laplacian_matrix(rp, ci, antisym, dt, material=mat, filt=(orc.FLUID, orc.ALL)) / rho0
            val = -theta * val
            for i in range(N * N):
                row = slice(rp[i], rp[i + 1])
                k = np.nonzero(ci[row] == i)[0][0]
                val[rp[i] + k] += 1.0
            vstar = np.zeros_like(v)
            ilu = orc.ILU(rp, ci, val, 0)
            for k in range(2):
                xs, info, _ = orc.solve(rp, ci, val, np.ascontiguousarray(rhs[:, k]), x0=np.ascontiguousarray(v[:, k]),
                                        singular=False, prec="ilu", ilu=ilu, params=prm)
                assert info.converged
                vstar[:, k] = xs
        vstar[:, 2] = 0.0
        # Poisson
        rp, ci, val, b = P.poisson(dt, rho, ghost(vstar), antisym=antisym, singular=orc.NULLSPACE)
        ilu = orc.ILU(rp, ci, val, 0)
        dp, info, _ = orc.solve(rp, ci, val, b, singular=True, prec="ilu", ilu=ilu, params=prm)
        assert info.converged
        if incremental:
            dp -= dp.mean()                          # pair_isph.cpp:1022
        vpre = vstar.copy()
        vstar = vstar - dt / rho0 * P.gradient(ghost(dp), antisym, filt=(orc.FLUID, orc.FLUID))
        vstar[:, 2] = 0.0
        p = p + dp if incremental else dp.copy()
        t = dt * step

        def err(xx, vv, pp):
            uex, pex = tgv_exact(xx, t, umax, nu, rho0)
            return dict(step=step, time=t, p_err=np.sqrt(np.mean((pp - pex - pp.mean()) ** 2)), p_norm=np.sqrt(np.mean(pex ** 2)),
                        u_err=np.sqrt(np.mean(np.sum((vv - uex) ** 2, axis=1))), u_norm=np.sqrt(np.mean(np.sum(uex ** 2, axis=1))))
        if not err_after:
            rec = err(x, vstar if err_on == "vstar" else v, p)
        # advanceTime
        if advance == "trap":
            dxp = 0.5 * dt * (vstar + v)
        elif advance == "new":
            dxp = dt * vstar
        elif advance == "trap_pre":
            dxp = 0.5 * dt * (vpre + v)
        elif advance == "new_pre":
            dxp = dt * vpre
        else:
            dxp = dt * v
        if padvect:
            gpn = P.gradient(ghost(p), antisym, filt=(orc.FLUID, orc.FLUID))
            p = p + np.sum(gpn * dxp, axis=1)
        x = x + dxp
        x[:, :2] %= L
        v = vstar
        if shift > 0.0:                               # fix isph/shift (fix_isph_shift.cpp:147-160, pair_isph_corrected.cpp:1203-1260)
            parts2, own2 = periodic_particles(x, L, cut)
            parts2["h"], parts2["cut"] = h, cut
            P2 = orc.Particles(parts2, own2, kernel=kernel)
            P2.precompute(corrections=corrections)
            vmax = np.sqrt(np.sum(v * v, axis=1)).max() if shift_v == 'max' else np.sqrt(np.sum(v * v, axis=1)).mean()
            dr = P2.compute_shift(shift * dt * vmax, cut * shiftcut_over_cut, 0.25)
            xs, vs, ps = P2.apply_shift(antisym, dr, np.ascontiguousarray(v[own2]), np.ascontiguousarray(p[own2]), sequential=shift_seq)
            x = xs[:N * N].copy()
            if not shift_nointerp:
                v, p = vs[:N * N].copy(), ps[:N * N].copy()
            x[:, :2] %= L
        if err_after:
            rec = err(x, v, p)
        out.append(rec)
    return out if all_steps else out[-1]


def score(rec, ref):
    return max(abs(rec[k] / ref[k] - 1.0) for k in ("p_err", "u_err", "p_norm", "u_norm"))


if __name__ == "__main__":
    kernel = sys.argv[1] if len(sys.argv) > 1 else "wendland"
    grid = dict(theta=[0.0, 0.5, 1.0], incremental=[True, False], antisym=[True, False], err_after=[False, True],
                advance=["trap", "new"], padvect=[True, False])
    keys = list(grid)
    rows = []
    for combo in itertools.product(*[grid[k] for k in keys]):
        kw = dict(zip(keys, combo))
        if not kw["incremental"] and kw["padvect"]:
            continue
        s = 0.0
        recs = {}
        for N in (16, 32):
            ref = REV390[kernel][N]
            recs[N] = run(N, ref["step"], kernel=kernel, **kw)
            s = max(s, score(recs[N], ref))
        rows.append((s, kw, recs))
        print("%.3e  %s  | N16 p %.4e u %.4e | N32 p %.4e u %.4e" % (s, kw, recs[16]["p_err"], recs[16]["u_err"], recs[32]["p_err"], recs[32]["u_err"]), flush=True)
    rows.sort(key=lambda r: r[0])
    print("\nbest:")
    for s, kw, recs in rows[:8]:
        print("%.3e %s" % (s, kw))
        for N in (16, 32):
            ref = REV390[kernel][N]
            print("   N=%d p_err %.9e (%.9e) u_err %.9e (%.9e) norms %.8e %.8e (%.8e %.8e)" % (
                N, recs[N]["p_err"], ref["p_err"], recs[N]["u_err"], ref["u_err"], recs[N]["p_norm"], recs[N]["u_norm"], ref["p_norm"], ref["u_norm"]))
