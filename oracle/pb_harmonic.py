"""TEST INFRASTRUCTURE (oracle side).  Second pin of the oracle against numbers the reference itself holds:
sph-script/conv-poisson-boltzmann-harmonic-2d-rev390.txt, produced by sph-script/poisson-boltzmann-harmonic-2d.lmp +
poisson-boltzmann-harmonic.xml through fix isph/error (fix_isph_error.cpp:188-345).

The run solves the manufactured Poisson-Boltzmann problem  F(psi) = -lap_h(psi) + kappa^2 sinh(psi) + f = 0  on the
periodic square [-pi, pi)^2 (lattice sq dx origin 0, h = 1.5 dx, Wendland cut 2h) with
kappa^2 = 2 ezcb / psiref = 1, f = -2 sin x cos y - sinh(sin x cos y), exact solution psi = sin x cos y
(functor_poisson_boltzmann_f.h:40-94, functor_poisson_boltzmann_extra_f.h).  lap_h is the corrected (Symmetric family)
SPH Laplacian with G_i / L_i: F uses the operator form (functor_laplacian.h:79-268), the Jacobian the matrix form
(functor_laplacian_matrix.h:73-316) -- the same linear operator, which is the one the pressure path assembles for
"Use Momentum Preserve Operator = Disabled".  The Newton driver (NOX) and Poisson-Boltzmann physics are outside the hot
path; here the nonlinear problem is solved to round-off with scipy so that the numbers below depend on nothing but the
oracle's kernel, volumes V_i, correction tensors G_i, L_i, Laplacian rows and corrected gradient:

    total volume        = sum_i V_i                                   (FunctorOuterVolume, functor_volume.h:40-80)
    err.psi.norm2       = sqrt( sum_i (psi_i - psi_exact)^2 / n )
    err.psi.grad.norm2  = sqrt( sum_i |grad_h psi_i - grad psi_exact|^2 / n )   (FunctorOuterGradient, Symmetric)

usage: python oracle/pb_harmonic.py [N ...]"""
import os
import sys

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, "..")]
import oracle as orc  # noqa: E402
from tgv_driver import periodic_particles  # noqa: E402

def known_answers():
    """the reference's own table rows (data): tests/golden/reference_known_answers.json"""
    import json
    g = json.load(open(os.path.join(_HERE, "..", "tests", "golden", "reference_known_answers.json")))
    return {int(k): v for k, v in g["conv_poisson_boltzmann_harmonic_2d_rev390"]["rows"].items()}


def lattice(N):
    """lattice sq dx origin 0 on [-pi, pi)^2, stored on [0, 2 pi) (the operator is translation invariant); returns the
    positions, the box-centred coordinates the analytic functions are evaluated at, h and the cut radius"""
    L = 2 * np.pi
    dx = L / N
    g = np.arange(N) * dx
    X, Y = np.meshgrid(g, g, indexing="xy")
    x = np.stack([X.ravel(), Y.ravel(), np.zeros(N * N)], axis=1)
    return x, x[:, 0] - np.pi, x[:, 1] - np.pi, 1.5 * dx, 3.0 * dx


def newton(apply_A, solve_J, rhs, n, max_it=30):
    """psi with A psi + sinh psi = rhs; stops when the update is at round-off level"""
    psi = np.zeros(n)
    for it in range(1, max_it + 1):
        F = apply_A(psi) + np.sinh(psi) - rhs
        d = solve_J(psi, F)
        psi -= d
        if np.max(np.abs(d)) <= 4e-16 * max(1.0, np.max(np.abs(psi))) * n ** 0.5:
            break
    return psi, it, float(np.max(np.abs(apply_A(psi) + np.sinh(psi) - rhs)))


def run(N, kernel="wendland"):
    x, xs, ys, h, cut = lattice(N)
    n = N * N
    parts, own = periodic_particles(x, 2 * np.pi, cut)
    parts["h"], parts["cut"] = h, cut
    P = orc.Particles(parts, own, kernel=kernel)
    P.precompute(corrections=True)
    nall = parts["nall"]
    exact = np.sin(xs) * np.cos(ys)
    gex = np.stack([np.cos(xs) * np.cos(ys), -np.sin(xs) * np.sin(ys)], axis=1)
    rp, ci = P.graph()
    val = P.laplacian_matrix(rp, ci, antisym=False, alpha=-1.0, material=np.ones(nall), filt=(orc.FLUID, orc.ALL))
    A = sps.csr_matrix((val, ci, rp), shape=(n, n))           # -lap_h
    rhs = 2.0 * exact + np.sinh(exact)                        # -f
    psi, it, res = newton(lambda p: A @ p, lambda p, F: spla.spsolve((A + sps.diags(np.cosh(p))).tocsc(), F), rhs, n)
    # the operator form the reference evaluates F with gives the same residual
    lap = P.laplacian_apply(np.ascontiguousarray(psi[own]), antisym=False, alpha=-1.0, material=np.ones(nall))[:, 0]
    op_vs_matrix = float(np.max(np.abs(lap - A @ psi)))
    grad = P.gradient(np.ascontiguousarray(psi[own]), antisym=False, alpha=1.0, filt=(orc.FLUID, orc.ALL))[:, :2]
    return dict(N=N, volume=float(P.vfrac[:n].sum()), err_psi=float(np.sqrt(np.sum((psi - exact) ** 2) / n)),
                err_grad=float(np.sqrt(np.sum((grad - gex) ** 2) / n)), sol_psi=float(np.sqrt(np.sum(exact ** 2) / n)),
                sol_grad=float(np.sqrt(np.sum(gex ** 2) / n)), newton_iterations=it, residual=res,
                operator_vs_matrix=op_vs_matrix)


if __name__ == "__main__":
    REV390 = known_answers()
    for N in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
        r = run(N)
        ref = REV390[N]
        print("N = %d   newton %d its, |F| %.1e, operator form vs matrix form %.1e" % (N, r["newton_iterations"], r["residual"], r["operator_vs_matrix"]))
        for name, key in (("total volume", "volume"), ("sol.psi.norm2", "sol_psi"), ("err.psi.norm2", "err_psi"),
                          ("sol.psi.grad.norm2", "sol_grad"), ("err.psi.grad.norm2", "err_grad")):
            print("    %-20s oracle %.15e   reference %.15e   rel. diff %.2e" % (name, r[key], ref[key], abs(r[key] - ref[key]) / ref[key]))
