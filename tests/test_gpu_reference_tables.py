"""-m gpu: the device path against numbers the REFERENCE holds (not the oracle):
sph-script/conv-poisson-boltzmann-harmonic-2d-rev390.txt, printed by fix isph/error (fix_isph_error.cpp:188-345) for
poisson-boltzmann-harmonic-2d.lmp.  The table depends on the kernel, V_i, G_i, L_i, the corrected (Symmetric-family)
Laplacian rows and the corrected gradient -- all computed on the device here: isph_compute_volumes,
isph_compute_corrections, isph_assemble_poisson (dt = 1, rho = 1, NotSingular: A = -lap_h, the Jacobian's matrix part,
functor_poisson_boltzmann_jacobian.h), isph_spmv for the residual, isph_mat_create_csr + isph_solve (FGMRES + SA-AMG)
for the Newton corrections, isph_gradient for grad psi.  Only the Newton loop and cosh/sinh live on the host
(Poisson-Boltzmann/NOX is outside the hot path)."""
import numpy as np
import pytest

from isph_amd import hip, workload
import pb_harmonic
import pb_channel
import oracle as orc

pytestmark = pytest.mark.gpu


def device_chain(ctx, N):
    spec = workload.TGVSpec(dim=2, ncell=(N, N), brick=(8, 8), origin=(0.0, 0.0), mode=workload.LATTICE)
    p = workload.make_tgv(spec)
    n, nall = p["nlocal"], p["nall"]
    assert n == N * N and abs(spec.h - 1.5 * 2 * np.pi / N) < 1e-15
    colmap = workload.single_rank_colmap(p)
    own = p["owner_index"].astype(np.int64)
    xs, ys = p["x"][:n, 0] - np.pi, p["x"][:n, 1] - np.pi          # the script's box is [-pi, pi)^2
    exact = np.sin(xs) * np.cos(ys)
    gex = np.stack([np.cos(xs) * np.cos(ys), -np.sin(xs) * np.sin(ys)], axis=1)
    vf = hip.compute_volumes(ctx, p, colmap)
    vfrac = np.ascontiguousarray(vf[own])
    Gc, Lc = hip.compute_corrections(ctx, p, colmap, vfrac)
    A, b = hip.assemble_poisson(ctx, p, colmap, 1.0, np.ones(nall), np.zeros((nall, 3)), antisym=False,
                                singular=hip.NOT_SINGULAR, vfrac=vfrac, Gc=Gc, Lc=Lc)
    assert np.max(np.abs(b)) == 0.0
    rp, ci, val = A.export_csr()
    diag = np.nonzero(ci == np.repeat(np.arange(n), np.diff(rp)))[0]
    assert len(diag) == n
    rhs = 2.0 * exact + np.sinh(exact)
    prm = hip.SolverParams(tol=1e-13, max_iters=400)
    its = []

    def solve_J(psi, F):
        jv = val.copy()
        jv[diag] += np.cosh(psi)
        J = hip.Matrix.from_csr(ctx, rp, ci, jv)
        M = hip.PrecondAMG(ctx, J, params=hip.AmgParams(block=512))
        d = np.zeros(n)
        info = hip.solve(ctx, J, F.copy(), d, prec=M, singular=False, params=prm)
        assert info.converged == 1
        its.append(info.iters)
        M.close()
        J.close()
        return d

    psi, nit, res = pb_harmonic.newton(lambda q: A.spmv(q), solve_J, rhs, n)
    grad = hip.gradient(ctx, p, colmap, np.ascontiguousarray(psi[own]), vfrac, antisym=False, Gc=Gc, filt=(workload.FLUID_KIND, 127))[:, :2]      # (Fluid, All), pair_isph_corrected.cpp:546-547
    return dict(volume=float(vf[:n].sum()), err_psi=float(np.sqrt(np.sum((psi - exact) ** 2) / n)),
                err_grad=float(np.sqrt(np.sum((grad - gex) ** 2) / n)), newton=nit, residual=res, gmres=its)


@pytest.mark.parametrize("N", [16, 32, 64, 128, 256, 512, 1024])
def test_device_chain_reproduces_reference_pb_harmonic_table(gpu_ctx, N):
    """N = 1024 is the table's last row: 1 048 576 particles."""
    ref = pb_harmonic.known_answers()[N]
    r = device_chain(gpu_ctx, N)
    assert ref["particles"] == N * N
    assert abs(r["volume"] - ref["volume"]) <= 2e-12 * ref["volume"]
    # the reference's own run stops Newton / Belos at its tolerances: its printed digits carry that (rel. 2e-10 at N=256
    # against the round-off-converged oracle).  The error itself shrinks like 1/N^2 (3.6e-6 at N = 1024) while psi keeps
    # the solver's absolute accuracy, so the digits that can agree go down with N: >= 8 up to N = 512, >= 7 at N = 1024
    tol = 1e-8 if N <= 512 else 1e-7
    assert abs(r["err_psi"] - ref["err_psi"]) <= tol * ref["err_psi"], (r, ref)
    assert abs(r["err_grad"] - ref["err_grad"]) <= tol * ref["err_grad"], (r, ref)


# ---------------------------------------------------------------- conv-channel-edl-potential-2d-morrisholmes-rev722.txt
def device_channel(ctx, N, boundary):
    """The linearised Poisson-Boltzmann channel problem  -lap_h psi + kappa^2 psi = 0, psi = 1 on the wall particles, IS a
    Helmholtz system of the hot path: (I - theta dt nu lap_h) psi = b with theta = 1, dt nu = 1 / kappa^2, b = 0 on the
    fluid rows and the wall value on the solid (identity) rows -- isph_assemble_helmholtz with the MorrisHolmes mirror
    (functor_boundary_morris_holmes.h:49-64) and isph_solve, nothing on the host but the error norm."""
    parts, own = pb_channel.channel(N)
    n, nall = parts["nlocal"], parts["nall"]
    colmap = own.astype(np.int32)
    kinds = pb_channel.KINDS
    vf = hip.compute_volumes(ctx, parts, colmap)
    vfrac = np.ascontiguousarray(vf[own])
    pnd = np.ascontiguousarray(hip.compute_pnd(ctx, parts, colmap, kinds=kinds)[own]) if boundary == "MorrisHolmes" else None
    Gc, Lc = hip.compute_corrections(ctx, parts, colmap, vfrac)
    solid = parts["type"] == 2
    vel = np.zeros((nall, 3))
    vel[solid, 0] = 1.0                                   # wall potential, carried by the identity rows
    H, b = hip.assemble_helmholtz(ctx, parts, colmap, 1.0 / pb_channel.KAPPA ** 2, 1.0, np.ones(nall), np.ones(nall),
                                  np.zeros(nall), np.zeros((nall, 3)), np.zeros(3), vel, antisym=False, incremental=True,
                                  vfrac=vfrac, Gc=Gc, Lc=Lc, kinds=kinds, pnd=pnd, morris_safe_coeff=0.0)
    rhs = np.ascontiguousarray(b[:n])
    assert np.array_equal(rhs, solid[:n].astype(float))
    M = hip.PrecondAMG(ctx, H, params=hip.AmgParams(block=512))
    psi = np.zeros(n)
    info = hip.solve(ctx, H, rhs.copy(), psi, prec=M, singular=False, params=hip.SolverParams(tol=1e-13, max_iters=500))
    assert info.converged == 1
    fl = ~solid[:n]
    ex = pb_channel.exact(parts["x"][:n, 1][fl])
    return dict(particles=int(fl.sum()), volume=float(vf[:n][fl].sum()), err_psi=float(np.sqrt(np.mean((psi[fl] - ex) ** 2))),
                wall=float(np.max(np.abs(psi[~fl] - 1.0))), iters=info.iters)


@pytest.mark.parametrize("boundary", ["MorrisHolmes", "ConstExtension"])
@pytest.mark.parametrize("N", [32, 64, 128, 256, 512, 1024])
def test_device_chain_reproduces_reference_channel_table(gpu_ctx, N, boundary):
    ref = pb_channel.known_answers(boundary)[N]
    r = device_channel(gpu_ctx, N, boundary)
    assert r["particles"] == ref["particles"] and r["wall"] <= 1e-12
    assert abs(r["volume"] - ref["volume"]) <= 2e-12 * ref["volume"]
    # the smallest error of the table (MorrisHolmes, N = 1024: 9.5e-6) is where the reference's own solver tolerance
    # shows first (1.4e-8 against the round-off-converged oracle)
    assert abs(r["err_psi"] - ref["err_psi"]) <= 1e-7 * ref["err_psi"], (r, ref)


def test_device_pnd_matches_oracle(gpu_ctx):
    """isph_compute_pnd vs the oracle's restatement of functor_normal.h:57-133 on the channel geometry"""
    parts, own = pb_channel.channel(64)
    n = parts["nlocal"]
    P = orc.Particles(parts, own, kernel="wendland", kinds=pb_channel.KINDS)
    po = P.compute_pnd()
    pg = hip.compute_pnd(gpu_ctx, parts, own.astype(np.int32), kinds=pb_channel.KINDS)
    assert np.max(np.abs(pg - po[:n])) <= 1e-13 * np.abs(po).max()
    P.precompute(corrections=False)
    xi = po[:n] * P.vfrac[:n]
    assert abs(xi[np.abs(parts["x"][:n, 1]) < 0.5].max() - 1.0) < 1e-13 and 0.5 < xi.min() < 1.0   # half a spacing from the wall: 0.81
