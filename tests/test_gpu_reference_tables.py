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
def device_channel(ctx, N, boundary, h_over_dx=1.2):
    """The linearised Poisson-Boltzmann channel problem  -lap_h psi + kappa^2 psi = 0, psi = 1 on the wall particles, IS a
    Helmholtz system of the hot path: (I - theta dt nu lap_h) psi = b with theta = 1, dt nu = 1 / kappa^2, b = 0 on the
    fluid rows and the wall value on the solid (identity) rows -- isph_assemble_helmholtz with the MorrisHolmes mirror
    (functor_boundary_morris_holmes.h:49-64) and isph_solve, nothing on the host but the error norm."""
    parts, own = pb_channel.channel(N, h_over_dx)
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


@pytest.mark.parametrize("N", [32, 64])
def test_device_chain_on_the_earlier_channel_table_with_h_102(gpu_ctx, N):
    """conv-channel-edl-potential-2d-morrisholmes-rev406.txt ("h = 1.02 dx"): volume to the table's digits on the device,
    err.psi to the 4 digits that revision shares with the current operators (tests/test_oracle.py, oracle/pb_channel.py)"""
    ref = pb_channel.known_answers("rev406")[N]
    r = device_channel(gpu_ctx, N, "MorrisHolmes", h_over_dx=1.02)
    assert r["particles"] == ref["particles"] and r["wall"] <= 1e-12
    assert abs(r["volume"] - ref["volume"]) <= 2e-12 * ref["volume"]
    assert abs(r["err_psi"] - ref["err_psi"]) <= 5e-4 * ref["err_psi"], (r, ref)


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


# ---------------------------------------------------------------- the reference's own kernel classes (oracle/_ref)
@pytest.mark.parametrize("kernel,knum,support", [("wendland", 0, 2.0), ("quintic", 1, 3.0), ("cubic", 2, 2.0)])
@pytest.mark.parametrize("dim", [2, 3])
def test_device_kernel_values_equal_the_reference_kernel_classes(gpu_ctx, kernel, knum, support, dim):
    """oracle/_ref/libisph_refkernels.so is the reference's KernelFuncWendland / Quintic / Cubic compiled from its own
    headers (built in the build container by oracle/build.py, shipped with the tree).  Device side: isolated pairs of
    particles -- FunctorOuterVolume gives V = 1 / (W(0) + W(r)), so W(r) comes back through isph_compute_volumes."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libisph_refkernels.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libisph_refkernels.so is built where /root/reference exists and travels with the tree; not here")
    ref = ctypes.CDLL(path)
    ref.ref_kernel_table.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]
    ref.ref_kernel_val.restype = ctypes.c_double
    ref.ref_kernel_val.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    h = 0.0942
    cut = support * h
    s = np.concatenate([np.random.default_rng(2).uniform(0.02, support, 500), np.arange(0.5, support, 0.5),
                        support + np.array([1e-9, 0.1, 0.4])])     # the last three: listed neighbours outside the cut
    r = np.ascontiguousarray(s * h)
    npair = len(r)
    x = np.zeros((2 * npair, 3))
    x[0::2, 0] = 10.0 * np.arange(npair)                              # pairs far apart from each other
    x[1::2, 0] = x[0::2, 0] + r
    ptr = np.arange(2 * npair + 1, dtype=np.int32)
    idx = np.arange(2 * npair, dtype=np.int32) ^ 1                    # each particle's only neighbour is its partner
    parts = dict(dim=dim, nlocal=2 * npair, nall=2 * npair, x=x, type=np.ones(2 * npair, np.int32), neigh_ptr=ptr,
                 neigh_idx=idx, h=h, cut=cut)
    colmap = np.arange(2 * npair, dtype=np.int32)
    vf = hip.compute_volumes(gpu_ctx, parts, colmap, kernel=kernel)
    w0 = ref.ref_kernel_val(knum, dim, 0.0, h)
    w, dw = np.zeros(npair), np.zeros(npair)
    rr = np.ascontiguousarray(x[1::2, 0] - x[0::2, 0])                # the distance the device sees
    ref.ref_kernel_table(knum, dim, h, npair, rr.ctypes.data, w.ctypes.data, dw.ctypes.data)
    inside = rr * rr < cut * cut
    assert inside.sum() >= 500 and (~inside).sum() == 3
    wdev = 1.0 / vf[0::2] - w0
    assert np.max(np.abs(wdev[inside] - w[inside])) <= 1e-15 * w0 * 4      # W(r) to round-off of 1/V - W(0)
    assert np.max(np.abs(1.0 / vf[0::2][~inside] - w0)) <= 1e-15 * w0 * 4  # outside the cut: the self term only
