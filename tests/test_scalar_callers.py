"""The two scalar callers of the solver objects besides the Navier-Stokes step (SURVEY section 8b "Callers"):
solute transport (pair_isph.cpp:811-835, functor_solute_transport.h) and the applied electric potential
(pair_isph.cpp:635-657, functor_applied_electric_potential.h).  Both build on the Laplacian rows with the reference's
FilterMatchBinary and the buffer particle kinds (pair_isph.h:113-124).

CPU part: the oracle's restatement against what the functors promise.  -m gpu part: the HIP rows against the oracle
(pattern exact, values 1e-12) and the solves against closed forms."""
import numpy as np
import pytest
import scipy.sparse as sps

from isph_amd import workload
import oracle as orc
from problems import Problem, tgv_spec

KINDS = [orc.FLUID, orc.BUFFER_DIRICHLET, orc.BUFFER_NEUMANN, orc.SOLID]   # types 1..4
SLAB = 1.2


def zone_types(parts):
    """x-slabs of buffer particles at both ends (thicker than the cut at these sizes), a solid block in the middle, fluid
    elsewhere; images follow their owners"""
    own = parts["owner_index"]
    x = parts["x"][:parts["nlocal"]] % (2 * np.pi)
    t = np.ones(parts["nlocal"], dtype=np.int32)
    t[x[:, 0] < SLAB] = 2
    t[x[:, 0] > 2 * np.pi - SLAB] = 3
    t[(np.abs(x[:, 0] - np.pi) < 0.5) & (np.abs(x[:, 1] - np.pi) < 0.9)] = 4
    return t[own]


def two_buffer_types(parts):
    """the same channel without the solid block"""
    own = parts["owner_index"]
    x = parts["x"][:parts["nlocal"]] % (2 * np.pi)
    t = np.ones(parts["nlocal"], dtype=np.int32)
    t[x[:, 0] < SLAB] = 2
    t[x[:, 0] > 2 * np.pi - SLAB] = 3
    return t[own]


def fields(pr):
    x = pr.parts["x"]
    conc = 1.0 + 0.3 * np.sin(x[:, 0]) * np.cos(2 * x[:, 1])
    phi = np.cos(x[:, 0]) + 0.2 * x[:, 1]
    sigma = 1.0 + 0.25 * np.cos(x[:, 1])
    return conc, phi, sigma                      # on every particle; the tests hand ghosts their owners' values


CASES = [dict(dim=2, n=20, mode=workload.JITTER), dict(dim=3, n=12, mode=workload.JITTER),
         dict(dim=2, n=16, mode=workload.LATTICE, kernel="quintic", cut_over_h=3.0)]


# ------------------------------------------------------------------------------------------------ CPU: the oracle
@pytest.mark.parametrize("antisym", [True, False])
@pytest.mark.parametrize("theta", [0.0, 0.5, 1.0])
def test_oracle_solute_transport_is_what_the_functor_promises(antisym, theta):
    pr = Problem(tgv_spec(dim=2, n=20, mode=workload.JITTER), antisym=antisym, kinds=KINDS, types=zone_types)
    p, n = pr.parts, pr.n
    conc, _, _ = fields(pr)
    own = p["owner_index"]
    conc = conc[:n][own]                                       # ghosts carry the owners' concentration (forward comm)
    dt, dcoeff = pr.spec.dt, 0.37
    rp, ci, val, b = pr.P.solute_transport(dt, theta, dcoeff, conc, antisym=antisym)
    A = sps.csr_matrix((val, ci, rp), shape=(n, n))
    kind = np.asarray(KINDS)[p["type"][:n] - 1]
    fluid = kind == orc.FLUID
    # the Laplacian the functor asks for: alpha = dt D, no material, FilterMatchBinary(Fluid, Fluid - BufferNeumann)
    lv = pr.P.laplacian_matrix(rp, ci, antisym, dt * dcoeff, None, filt=(orc.FLUID | orc.FILTER_MATCH, orc.FLUID - orc.BUFFER_NEUMANN))
    L = sps.csr_matrix((lv, ci, rp), shape=(n, n))
    assert abs(L[~fluid]).sum() == 0.0                         # only rows of kind == Fluid are assembled
    solid_cols = np.flatnonzero(kind == orc.SOLID)
    assert abs(L[:, solid_cols]).sum() == 0.0                  # no coupling to solid neighbours ...
    for bk in (orc.BUFFER_DIRICHLET, orc.BUFFER_NEUMANN):      # ... but to both buffers (the mask is for Solid neighbours only)
        assert abs(L[fluid][:, np.flatnonzero(kind == bk)]).sum() > 0.0
    want = sps.eye(n) - theta * L
    want = want.tolil()
    for i in np.flatnonzero(~fluid):
        want[i, i] = 1.0
    assert abs(A - want.tocsr()).max() <= 1e-15 * max(1.0, abs(L).max())
    bw = conc[:n].copy()
    bw[fluid] += (1.0 - theta) * (L @ conc[:n])[fluid]
    assert np.max(np.abs(b - bw)) <= 1e-13 * np.abs(bw).max()
    assert np.array_equal(b[~fluid], conc[:n][~fluid])         # Dirichlet rows hand the old value through


@pytest.mark.parametrize("antisym", [True, False])
def test_oracle_applied_potential_is_what_the_functor_promises(antisym):
    pr = Problem(tgv_spec(dim=2, n=20, mode=workload.JITTER), antisym=antisym, kinds=KINDS, types=zone_types)
    p, n = pr.parts, pr.n
    _, phi, sigma = fields(pr)
    own = p["owner_index"]
    phi, sigma = phi[:n][own], sigma[:n][own]
    rp, ci, val, b = pr.P.applied_potential(sigma, phi, antisym=antisym)
    A = sps.csr_matrix((val, ci, rp), shape=(n, n))
    kind = np.asarray(KINDS)[p["type"][:n] - 1]
    fluid, buf = kind == orc.FLUID, (kind == orc.BUFFER_DIRICHLET) | (kind == orc.BUFFER_NEUMANN)
    lv = pr.P.laplacian_matrix(rp, ci, antisym, -1.0, sigma, filt=(orc.FLUID | orc.FILTER_MATCH, orc.FLUID))
    L = sps.csr_matrix((lv, ci, rp), shape=(n, n))
    assert abs((A - L)[fluid]).max() == 0.0                    # fluid rows: the Laplacian as assembled
    rest = A[~fluid]
    assert np.array_equal(rest.data[rest.data != 0.0], np.ones((~fluid).sum()))   # unit rows
    assert np.array_equal(A.diagonal()[~fluid], np.ones((~fluid).sum()))
    assert np.array_equal(b[buf], phi[:n][buf]) and not b[~buf].any()
    if antisym:                                                # conservative form: zero row sums on rows away from the solid
        far = fluid & (np.asarray(abs(L[:, np.flatnonzero(kind == orc.SOLID)]).sum(axis=1)).ravel() == 0)
        d = A.diagonal()
        near_solid = np.zeros(n, bool)
        Pat = sps.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
        near_solid[np.asarray(Pat[:, np.flatnonzero(kind == orc.SOLID)].sum(axis=1)).ravel() > 0] = True
        rows = far & ~near_solid
        assert rows.sum() > 20
        assert np.max(np.abs(np.asarray(A[rows].sum(axis=1)).ravel())) <= 1e-12 * d[rows].max()


# ------------------------------------------------------------------------------------------------ GPU: parity
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("antisym", [True, False])
@pytest.mark.parametrize("theta", [0.0, 0.5, 1.0])
def test_gpu_solute_transport_rows_match_oracle(gpu_ctx_both, case, antisym, theta):
    gpu_ctx = gpu_ctx_both
    from isph_amd import hip
    pr = Problem(tgv_spec(**case), antisym=antisym, kinds=KINDS, types=zone_types)
    p, n = pr.parts, pr.n
    conc, _, _ = fields(pr)
    conc = conc[:n][p["owner_index"]]
    rp, ci, val, b = pr.P.solute_transport(pr.spec.dt, theta, 0.37, conc, antisym=antisym)
    A, bg = hip.assemble_solute_transport(gpu_ctx, p, pr.colmap, pr.spec.dt, theta, 0.37, conc, antisym=antisym,
                                          vfrac=pr.P.vfrac, Gc=None if antisym else pr.P.Gc, Lc=None if antisym else pr.P.Lc,
                                          kernel=pr.spec.kernel, kinds=KINDS)
    rp2, ci2, v2 = A.export_csr()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci)             # pattern: exact
    assert np.max(np.abs(v2 - val)) <= 1e-12 * max(np.abs(val).max(), 1.0)
    assert np.max(np.abs(bg - b)) <= 1e-12 * np.abs(b).max()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("antisym", [True, False])
@pytest.mark.parametrize("with_sigma", [True, False])
def test_gpu_applied_potential_rows_match_oracle(gpu_ctx_both, case, antisym, with_sigma):
    gpu_ctx = gpu_ctx_both
    from isph_amd import hip
    pr = Problem(tgv_spec(**case), antisym=antisym, kinds=KINDS, types=zone_types)
    p, n = pr.parts, pr.n
    _, phi, sigma = fields(pr)
    own = p["owner_index"]
    phi, sigma = phi[:n][own], (sigma[:n][own] if with_sigma else None)
    rp, ci, val, b = pr.P.applied_potential(sigma, phi, antisym=antisym)
    A, bg = hip.assemble_applied_potential(gpu_ctx, p, pr.colmap, sigma, phi, antisym=antisym, vfrac=pr.P.vfrac,
                                           Gc=None if antisym else pr.P.Gc, Lc=None if antisym else pr.P.Lc,
                                           kernel=pr.spec.kernel, kinds=KINDS)
    rp2, ci2, v2 = A.export_csr()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci)
    assert np.max(np.abs(v2 - val)) <= 1e-12 * max(np.abs(val).max(), 1.0)
    assert np.array_equal(bg, b)                                           # copies of phi and zeros


@pytest.mark.gpu
@pytest.mark.parametrize("dim,n", [(2, 32), (3, 16)])
def test_gpu_applied_potential_between_two_buffers_is_linear(gpu_ctx, dim, n):
    """computeAppliedElectricField on a lattice channel: buffers at both ends hold phi = x, uniform conductivity: the
    discrete Laplace equation is satisfied by the linear potential exactly (symmetric stencil), so the solve (the
    reference's GMRES + ILU, solveProblem(prec, "AppliedElectricPotential")) must return phi = x on the fluid rows."""
    from isph_amd import hip
    pr = Problem(tgv_spec(dim=dim, n=n, mode=workload.LATTICE), kinds=KINDS, types=two_buffer_types)
    p, nl = pr.parts, pr.n
    own = p["owner_index"]
    xs = (p["x"][:nl, 0] % (2 * np.pi))[own]
    kind = np.asarray(KINDS)[p["type"][:nl] - 1]
    phi0 = np.where(kind[own] == orc.FLUID, 0.0, xs)                         # initial guess 0 in the fluid, data in the buffers
    A, b = hip.assemble_applied_potential(gpu_ctx, p, pr.colmap, None, phi0, vfrac=pr.P.vfrac, kinds=KINDS)
    M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 256)
    x = phi0[:nl].copy()
    info = hip.solve(gpu_ctx, A, b, x, prec=M, params=hip.SolverParams(tol=1e-12))
    assert info.converged == 1
    assert np.max(np.abs(x - xs[:nl])) <= 1e-9 * 2 * np.pi
    # the oracle's rows solved by the oracle's solver: the same potential
    rp, ci, val, bo = pr.P.applied_potential(None, phi0)
    xo, io, _ = orc.solve(rp, ci, val, bo, x0=phi0[:nl], prec="ilu", ilu=orc.ILU(rp, ci, val, 0), params=orc.SolverParams(tol=1e-12))
    assert io.converged == 1 and np.max(np.abs(x - xo)) <= 1e-9 * 2 * np.pi


@pytest.mark.gpu
def test_gpu_solute_transport_step_matches_oracle_solve_and_conserves_a_uniform_field(gpu_ctx_both):
    gpu_ctx = gpu_ctx_both
    """computeSoluteTransport: one theta = 0.5 step.  The solve equals the oracle's solve of the oracle's rows; a uniform
    concentration is a fixed point (zero row sums of the Laplacian on the Fluid rows, unit rows elsewhere)."""
    from isph_amd import hip
    pr = Problem(tgv_spec(dim=3, n=12, mode=workload.JITTER), kinds=KINDS, types=two_buffer_types)
    p, n = pr.parts, pr.n
    own = p["owner_index"]
    conc, _, _ = fields(pr)
    conc = conc[:n][own]
    dt, dcoeff = pr.spec.dt, 0.8
    A, b = hip.assemble_solute_transport(gpu_ctx, p, pr.colmap, dt, 0.5, dcoeff, conc, vfrac=pr.P.vfrac, kinds=KINDS)
    M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 256)
    x = conc[:n].copy()
    info = hip.solve(gpu_ctx, A, b, x, prec=M)
    rp, ci, val, bo = pr.P.solute_transport(dt, 0.5, dcoeff, conc)
    bp = np.arange(0, n + 256, 256).clip(0, n).astype(np.int32)
    xo, io, _ = orc.solve(rp, ci, val, bo, x0=conc[:n], prec="ilu", ilu=orc.ILU(rp, ci, val, 0, bp))
    assert info.converged == 1 and io.converged == 1 and abs(info.iters - io.iters) <= 1
    assert np.linalg.norm(x - xo) <= 1e-7 * np.linalg.norm(xo)
    one = np.ones(p["nall"])
    A1, b1 = hip.assemble_solute_transport(gpu_ctx, p, pr.colmap, dt, 0.5, dcoeff, one, vfrac=pr.P.vfrac, kinds=KINDS)
    y = A1.spmv(one[:n].copy())
    assert np.max(np.abs(y - 1.0)) <= 1e-12 and np.max(np.abs(b1 - 1.0)) <= 1e-12


@pytest.mark.gpu
def test_scalar_callers_reject_kinds_outside_the_functors_switch(gpu_ctx):
    from isph_amd import hip
    pr = Problem(tgv_spec(dim=2, n=12, mode=workload.JITTER))
    p = pr.parts
    with pytest.raises(hip.IsphError):
        hip.assemble_solute_transport(gpu_ctx, p, pr.colmap, 0.1, 0.5, 1.0, np.ones(p["nall"]), vfrac=pr.P.vfrac, kinds=[16])
