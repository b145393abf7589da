"""CPU checks of the oracle itself (-m "not gpu"): the one reference-functor
output recorded in SURVEY.md, analytic invariants of the SPH operators, the
solver restatement against SciPy."""
import numpy as np
import pytest
import scipy.sparse as sps
import scipy.sparse.linalg as spla

from isph_amd import workload
import oracle as orc
from problems import Problem, tgv_spec, wall_types


def _golden(name):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    return json.load(open(path)) if name.endswith(".json") else np.load(path)


def test_reference_probe_row_from_survey():
    """the reference's own functor
    (functor_laplacian_matrix.h, AntiSymmetric, Wendland, 2-D 8x8 periodic lattice,
    alpha=-0.01, material=1, V=dx^2) gave diag 3.6692e-02 and row-sum ~3e-18."""
    sp = tgv_spec(dim=2, n=8, mode=workload.LATTICE, brick=0)
    p = workload.make_tgv(sp)
    P = orc.Particles(p, workload.single_rank_colmap(p))
    P.vfrac[:] = sp.dx ** 2
    rp, ci = P.graph()
    val = P.laplacian_matrix(rp, ci, True, -0.01, material=np.ones(p["nall"]), filt=(orc.FLUID, orc.FLUID))
    row = slice(rp[0], rp[1])
    diag = val[row][ci[row] == 0][0]
    ref = _golden("reference_known_answers.json")["laplacian_row_probe"]
    assert "%.4e" % diag == "%.4e" % ref["diag"]
    assert abs(val[row].sum()) < ref["abs_row_sum_below"]
    assert np.count_nonzero(np.abs(val[row]) > 1e-12) == ref["nnz"]


@pytest.mark.parametrize("kernel,cut", [("wendland", 2.0), ("quintic", 3.0), ("cubic", 2.0)])
@pytest.mark.parametrize("dim", [2, 3])
def test_kernel_normalisation_and_derivative(kernel, cut, dim):
    h = 0.37
    r = np.linspace(0, cut * h, 4001)
    w = np.array([orc.kernel_val(kernel, dim, x, h) for x in r])
    shell = 2 * np.pi * r if dim == 2 else 4 * np.pi * r ** 2
    # unit integral -- except the reference's 3-D quintic constant 14/(1745 pi h^3)
    # (kernel_quintic.h:43), which integrates to 0.96275; the oracle follows the reference.
    want = 0.962751 if (kernel == "quintic" and dim == 3) else 1.0
    assert abs(np.trapezoid(w * shell, r) - want) < 2e-4
    dw = np.array([orc.kernel_dval(kernel, dim, x, h) for x in r])
    num = np.gradient(w, r)
    assert np.max(np.abs(dw[2:-2] - num[2:-2])) < 2e-3 * np.max(np.abs(dw))
    assert orc.kernel_val(kernel, dim, cut * h * 1.0001, h) == 0.0


def test_volume_on_lattice_is_cell_volume():
    pr = Problem(tgv_spec(dim=3, n=8, mode=workload.LATTICE))
    assert np.allclose(pr.P.vfrac[:pr.n], pr.spec.dx ** 3, rtol=2e-2)
    assert np.ptp(pr.P.vfrac[:pr.n]) < 1e-12


def test_corrections_on_lattice():
    """Symmetric family tensors on a regular lattice: G_i ~ I, L_i ~ I."""
    pr = Problem(tgv_spec(dim=2, n=12, mode=workload.LATTICE), antisym=False)
    G = pr.P.Gc[:pr.n].reshape(-1, 2, 2)
    assert np.allclose(G, np.eye(2) * G[0, 0, 0], atol=1e-10)
    assert abs(G[0, 0, 0] - 1.0) < 5e-2
    L = pr.P.Lc[:pr.n]
    assert np.allclose(L[:, 1], 0, atol=1e-10) and np.allclose(L[:, 0], L[:, 2], rtol=1e-10)


@pytest.mark.parametrize("antisym", [True, False])
@pytest.mark.parametrize("dim,n", [(2, 16), (3, 10)])
def test_poisson_matrix_invariants(antisym, dim, n):
    pr = Problem(tgv_spec(dim=dim, n=n, mode=workload.JITTER), antisym=antisym)
    rp, ci, val, b = pr.poisson()
    A = sps.csr_matrix((val, ci, rp), shape=(pr.n, pr.n))
    scale = abs(A.diagonal()).max()
    assert abs(A @ np.ones(pr.n)).max() < 1e-12 * scale        # constants in the null space
    assert (A.diagonal() > 0).all()                             # alpha=-dt: positive diagonal
    if antisym:
        sym_err = abs(A - A.T).max() / scale
        assert sym_err < 5e-2                                   # symmetric up to the grad(m) term
    # consistency: A p ~ -dt/rho * laplacian(p) for a smooth p
    x = pr.parts["x"][:pr.n]
    pfun = np.cos(x[:, 0]) * np.cos(x[:, 1])
    lap = -2.0 * pfun
    got = A @ pfun
    want = -pr.spec.dt / pr.spec.rho * lap
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert err < ((0.3 if dim == 3 else 0.25) if antisym else (0.15 if dim == 3 else 0.05)), err


def test_divergence_and_gradient_consistency():
    pr = Problem(tgv_spec(dim=2, n=24, mode=workload.JITTER), antisym=False)
    x = pr.parts["x"]
    f = np.stack([np.sin(x[:, 0]), np.cos(x[:, 1]), np.zeros(len(x))], axis=1)
    div = pr.P.divergence(f, antisym=False)
    want = np.cos(x[:pr.n, 0]) - np.sin(x[:pr.n, 1])
    assert np.linalg.norm(div - want) / np.linalg.norm(want) < 0.03
    g = pr.P.gradient(np.sin(x[:, 0]) * np.cos(x[:, 1]), antisym=False)
    wx = np.cos(x[:pr.n, 0]) * np.cos(x[:pr.n, 1])
    assert np.linalg.norm(g[:, 0] - wx) / np.linalg.norm(wx) < 0.06


def test_duplicate_images_merge_like_fillcomplete():
    """8x8 box is narrower than 2*cut+...: periodic images share a tag and the
    graph must merge them (functor_graph.h:92-98)."""
    pr = Problem(tgv_spec(dim=2, n=4, mode=workload.JITTER, brick=0))
    rp, ci, val, b = pr.poisson()
    for i in range(pr.n):
        row = ci[rp[i]:rp[i + 1]]
        assert len(np.unique(row)) == len(row) and (np.diff(row) > 0).all()
    assert np.diff(pr.parts["neigh_ptr"]).max() + 1 > np.diff(rp).max()


@pytest.mark.parametrize("mode", [orc.PINZERO, orc.DOUBLEDIAG])
def test_singular_modes(mode):
    base = Problem(tgv_spec(dim=2, n=12, mode=workload.JITTER))
    rp, ci, v0, b0 = base.poisson()
    pr = Problem(tgv_spec(dim=2, n=12, mode=workload.JITTER), singular=mode)
    rp1, ci1, v1, b1 = pr.poisson()
    A0 = sps.csr_matrix((v0, ci, rp)).toarray()
    A1 = sps.csr_matrix((v1, ci1, rp1)).toarray()
    if mode == orc.PINZERO:
        assert A1[0, 0] == -1.0 and np.count_nonzero(A1[0]) == 1 and b1[0] == 0.0
    else:
        assert np.isclose(A1[0, 0], 1.5 * A0[0, 0])
    assert np.allclose(A1[1:], A0[1:])


def _aug_solve(A, b, nv):
    n = A.shape[0]
    bp = b - nv * (nv @ b)
    aug = sps.bmat([[A, nv[:, None]], [nv[None, :], None]]).tocsc()
    return spla.spsolve(aug, np.concatenate([bp, [0.0]]))[:n]


@pytest.mark.parametrize("prec,lof,bs", [("none", 0, 0), ("jacobi", 0, 0), ("ilu", 0, 256), ("ilu", 0, 0), ("ilu", 1, 0)])
def test_solver_restatement_vs_scipy(prec, lof, bs):
    pr = Problem(tgv_spec(dim=3, n=12, mode=workload.JITTER))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    A = sps.csr_matrix((val, ci, rp), shape=(n, n))
    ilu = None
    if prec == "ilu":
        bp = None if bs == 0 else np.arange(0, n + bs, bs).clip(0, n)
        ilu = orc.ILU(rp, ci, val, lof, bp)
    x, info, bproj = orc.solve(rp, ci, val, b, singular=True, prec=prec, ilu=ilu)
    assert info.converged and info.rel_res_implicit <= 1e-8
    nv = np.ones(n) / np.sqrt(n)
    assert abs(x @ nv) < 1e-12 * np.linalg.norm(x)
    xs = _aug_solve(A, b, nv)
    assert np.linalg.norm(x - xs) / np.linalg.norm(xs) < 1e-6
    r = bproj - A @ x
    r -= nv * (nv @ r)
    assert np.linalg.norm(r) / np.linalg.norm(bproj) < 2e-8


def test_ilu0_matches_dense_definition():
    pr = Problem(tgv_spec(dim=2, n=10, mode=workload.JITTER))
    rp, ci, val, _ = pr.poisson()
    n = pr.n
    A = sps.csr_matrix((val, ci, rp), shape=(n, n)).toarray()
    pat = A != 0
    LU = A.copy()
    for i in range(1, n):                       # textbook IKJ ILU(0)
        for k in range(i):
            if pat[i, k]:
                LU[i, k] /= LU[k, k]
                for j in range(k + 1, n):
                    if pat[i, j]:
                        LU[i, j] -= LU[i, k] * LU[k, j]
    frp, fci, fv = orc.ILU(rp, ci, val, 0).export()
    F = sps.csr_matrix((fv, fci, frp), shape=(n, n)).toarray()
    assert np.allclose(F[pat], LU[pat], rtol=1e-12, atol=1e-14)
    # apply == U^-1 L^-1 r
    r = np.random.default_rng(0).standard_normal(n)
    Lm = np.tril(LU, -1) + np.eye(n)
    Um = np.triu(LU)
    z = np.linalg.solve(Um, np.linalg.solve(Lm, r))
    assert np.allclose(orc.ILU(rp, ci, val, 0).apply(r), z, rtol=1e-10)


@pytest.mark.parametrize("k", [1, 2, 3])
def test_iluk_matches_dense_definition(k):
    """ILU(k) as Ifpack builds it: level-of-fill graph first (lev(i,j) = min_p lev(i,p)+lev(p,j)+1 <= k, entries of A at
    level 0), then IKJ elimination restricted to that FINAL pattern (Ifpack_IlukGraph + Ifpack_ILU::Compute)."""
    pr = Problem(tgv_spec(dim=2, n=8, mode=workload.JITTER))
    rp, ci, val, _ = pr.poisson()
    n = pr.n
    A = sps.csr_matrix((val, ci, rp), shape=(n, n)).toarray()
    INF = 10 ** 6
    lev = np.where(A != 0, 0, INF)
    lev[np.arange(n), np.arange(n)] = 0
    for i in range(n):                          # textbook symbolic phase, rows in order
        for p in range(i):
            if lev[i, p] <= k:
                for j in range(p + 1, n):
                    if lev[p, j] <= k:
                        lev[i, j] = min(lev[i, j], lev[i, p] + lev[p, j] + 1)
    pat = lev <= k
    LU = A.copy()
    for i in range(1, n):
        for p in range(i):
            if pat[i, p]:
                LU[i, p] /= LU[p, p]
                for j in range(p + 1, n):
                    if pat[i, j]:
                        LU[i, j] -= LU[i, p] * LU[p, j]
    frp, fci, fv = orc.ILU(rp, ci, val, k).export()
    got = np.zeros((n, n), bool)
    for i in range(n):
        got[i, fci[frp[i]:frp[i + 1]]] = True
    assert np.array_equal(got, pat) and pat.sum() > (A != 0).sum()
    F = sps.csr_matrix((fv, fci, frp), shape=(n, n)).toarray()
    assert np.allclose(F[pat], LU[pat], rtol=1e-11, atol=1e-13)


def test_cg_on_symmetric_lattice_system():
    """CG + ILU(0): the USER-REAXC-T / config-1 setting (Block CG, tol 1e-6)."""
    pr = Problem(tgv_spec(dim=2, n=32, mode=workload.LATTICE))
    rp, ci, val, _ = pr.poisson()
    x = pr.parts["x"][:pr.n]
    b = np.cos(2 * x[:, 0]) + np.cos(2 * x[:, 1])
    ilu = orc.ILU(rp, ci, val, 0)
    xs, info, bp = orc.solve(rp, ci, val, b, singular=True, prec="ilu", ilu=ilu,
                             params=orc.SolverParams(solver_type=1, tol=1e-6))
    assert info.converged and info.iters < 60
    A = sps.csr_matrix((val, ci, rp))
    r = bp - A @ xs
    assert np.linalg.norm(r - r.mean()) / np.linalg.norm(bp) < 1e-5


def _rel(a, b):
    return abs(a / b - 1.0)


@pytest.mark.parametrize("kernel,N", [("wendland", 16), ("wendland", 32), ("wendland", 64), ("wendland", 128),
                                      ("quintic", 16), ("quintic", 32), ("quintic", 64)])
def test_tgv2d_known_answer_table_pinned(kernel, N):
    """PIN of the oracle against numbers the reference itself recorded:
    sph-script/conv-taylor-green-vortex-2d-rev390.txt (Wendland :6-29, Quintic :41-64) and ...-rev230.txt,
    produced by fix_isph_tgv.cpp:43-125.  The oracle's computePre + Helmholtz (theta = 1/2) + Poisson (NullSpace,
    FGMRES + ILU(0)) + corrections + advanceTime, chained by oracle/tgv_driver.py with the one combination of
    unrecorded settings that fits (oracle/tgv_sweep.py: theta 1/2, incremental pressure, Symmetric corrected
    operators, error on vstar before advanceTime), reproduces BOTH printed error columns of every row to 3
    significant digits (<= 2.5e-3 relative; Quintic N=16 pressure 5.3e-3)."""
    import tgv_driver as T
    gold = _golden("reference_known_answers.json")
    key = "conv_taylor_green_vortex_2d_rev390" + ("" if kernel == "wendland" else "_quintic")
    ref = gold[key]["rows"][str(N)]
    ref230 = gold[key.replace("rev390", "rev230")]["rows"][str(N)]
    h = T.run_tgv2d(N, ref["step"], kernel=kernel, **T.PINNED)[-1]
    assert h["step"] == ref["step"] and abs(h["time"] - ref["time"]) < 1e-6
    tol_p = 6e-3 if (kernel, N) == ("quintic", 16) else 2.5e-3
    assert _rel(h["p_err"], ref["p_err"]) < tol_p, (h["p_err"], ref["p_err"])
    assert _rel(h["u_err"], ref["u_err"]) < 2.5e-3, (h["u_err"], ref["u_err"])
    assert _rel(h["p_err"], ref230["p_err"]) < tol_p and _rel(h["u_err"], ref230["u_err"]) < 2.5e-3
    # the norms in parentheses only depend on the particle positions: without the shift they sit 4e-4 off
    assert _rel(h["p_norm"], ref["p_norm"]) < 1e-3 and _rel(h["u_norm"], ref["u_norm"]) < 1e-3


@pytest.mark.parametrize("N", [32, 64])
def test_tgv2d_known_answer_table_with_shift(N):
    """Same rows with `fix isph/shift 0.05` of the script (fix_isph_shift.cpp:147-160) scaled by the mean fluid
    speed (the alternative commented at pair_isph_corrected.cpp:1235): the position-only norms the table prints
    then agree to <= 1e-4 (4e-4 without shift: the reference run did shift its particles) and both error
    columns to <= 4e-4, i.e. 3.5+ significant digits."""
    import tgv_driver as T
    ref = _golden("reference_known_answers.json")["conv_taylor_green_vortex_2d_rev390"]["rows"][str(N)]
    h = T.run_tgv2d(N, ref["step"], **T.PINNED, **T.PINNED_SHIFT)[-1]
    assert _rel(h["p_err"], ref["p_err"]) < 4e-4 and _rel(h["u_err"], ref["u_err"]) < 4e-4
    assert _rel(h["p_norm"], ref["p_norm"]) < 1e-4 and _rel(h["u_norm"], ref["u_norm"]) < 1e-4


def test_tgv2d_other_settings_do_not_fit_the_table():
    """The pin is discriminating: the settings of today's xml (theta = 0, taylor-green-vortex.xml:15) or the
    AntiSymmetric family (today's default, pair_isph.cpp:1779) miss the table by >= 10 %."""
    import tgv_driver as T
    ref = _golden("reference_known_answers.json")["conv_taylor_green_vortex_2d_rev390"]["rows"]["16"]
    for kw in (dict(theta=0.0, incremental=True, antisym=False), dict(theta=0.5, incremental=True, antisym=True),
               dict(theta=0.5, incremental=False, antisym=False)):
        h = T.run_tgv2d(16, ref["step"], **kw)[-1]
        assert max(_rel(h["p_err"], ref["p_err"]), _rel(h["u_err"], ref["u_err"])) > 0.1, kw


def test_shift_serial_in_place_and_pre_shift_state_differ_at_second_order():
    """The reference applies the shift in a serial in-place loop (pair_for.h:9-14, functor_apply_shift.h:76-108);
    a parallel device reads the pre-shift state.  The oracle restates both: the gap must scale like |dr|^2."""
    pr = Problem(tgv_spec(dim=2, n=16, mode=workload.JITTER))
    parts, P = pr.parts, pr.P
    nl, own = parts["nlocal"], parts["owner_index"]
    v = np.ascontiguousarray(parts["v"][:nl][own])
    p = np.cos(parts["x"][:, 0]) * np.sin(parts["x"][:, 1])
    gaps = []
    for alpha in (1e-3, 1e-4):
        dr = P.compute_shift(alpha, parts["cut"], 0.0)
        xs, vs, ps = P.apply_shift(True, dr, v, p, sequential=True)
        xj, vj, pj = P.apply_shift(True, dr, v, p, sequential=False)
        assert np.array_equal(xs, xj)                       # positions do not depend on the order
        first = np.abs(pj - p).max()
        gaps.append(np.abs(ps - pj).max() / first)
        assert first > 0 and gaps[-1] < 0.2
    assert gaps[1] < 0.2 * gaps[0]                          # relative gap ~ |dr|


# ---------------------------------------------------------------- smoothed-aggregation AMG restatement
def test_amg_oracle_hierarchy_invariants_and_convergence():
    """oracle/isph_amg_oracle.c: aggregates partition the nodes, P reproduces the null vector, the Galerkin
    operator keeps it in its kernel, and the cycle beats point Jacobi as a GMRES preconditioner."""
    pr = Problem(tgv_spec(dim=3, n=20, mode=workload.JITTER, brick=4))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    nv = np.ones(n) / np.sqrt(n)
    G = orc.AMG(rp, ci, val, nullvec=nv, theta=0.02, block=256, coarse_max=64)
    assert G.levels >= 2
    agg = G.aggregates(0)
    n1 = G.level_info(1)["rows"]
    assert agg.min() == 0 and agg.max() == n1 - 1 and len(np.unique(agg)) == n1
    rP, cP, vP = G.export(0, "P")
    P = sps.csr_matrix((vP, cP, rP), shape=(n, n1))
    A = sps.csr_matrix((val, ci, rp), shape=(n, n))
    nc = np.sqrt(np.bincount(agg, weights=nv * nv))          # coarse null vector = aggregate norms
    assert np.linalg.norm(P @ nc - nv) <= 1e-12              # (I - w D^-1 A) P_t n_c = n - w D^-1 A n = n
    r1, c1, v1 = G.export(1, "A")
    A1 = sps.csr_matrix((v1, c1, r1), shape=(n1, n1))
    assert abs(A1 - P.T @ A @ P).max() <= 1e-12 * abs(A1).max()
    assert np.linalg.norm(A1 @ nc) <= 1e-10 * abs(A1).max()
    x, info, _ = orc.solve(rp, ci, val, b, singular=True, prec="amg", amg=G)
    xj, infoj, _ = orc.solve(rp, ci, val, b, singular=True, prec="jacobi")
    assert info.converged == 1 and info.iters < infoj.iters
    assert np.linalg.norm(x - xj) <= 1e-6 * np.linalg.norm(xj)


def test_amg_oracle_nonsingular_direct_coarse_solve():
    pr = Problem(tgv_spec(dim=2, n=40, mode=workload.JITTER, brick=8), singular=orc.NOT_SINGULAR,
                 kinds=[orc.FLUID, orc.SOLID], types=wall_types)
    rp, ci, val, b = pr.poisson()
    G = orc.AMG(rp, ci, val, theta=0.05, block=256, coarse_max=64)
    x, info, _ = orc.solve(rp, ci, val, b, singular=False, prec="amg", amg=G)
    A = sps.csr_matrix((val, ci, rp), shape=(pr.n, pr.n))
    assert info.converged == 1
    assert np.linalg.norm(b - A @ x) <= 2e-8 * np.linalg.norm(b)


def test_oracle_reproduces_committed_golden_fixture():
    """tests/golden/tgv2d_walls_12.npz (made by tests/golden/make_golden.py): the oracle of today must still produce
    the matrices, right-hand sides, ILU(0) factor, solution and aggregates that were committed."""
    sys_path_golden = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", __import__("os").path.join(sys_path_golden, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    G = _golden("tgv2d_walls_12.npz")
    prs = mg.golden_problem()
    for fam, pr in prs.items():
        rp, ci, val, b = pr.poisson()
        assert np.array_equal(rp, G[fam + "_rowptr"]) and np.array_equal(ci, G[fam + "_colidx"])
        assert np.max(np.abs(val - G[fam + "_val"])) <= 1e-13 * np.abs(val).max()
        assert np.max(np.abs(b - G[fam + "_b"])) <= 1e-13 * np.abs(b).max()
    pr = prs["antisym"]
    rp, ci, val, b = pr.poisson()
    bp = np.arange(0, pr.n + 64, 64).clip(0, pr.n).astype(np.int32)
    ilu = orc.ILU(rp, ci, val, 0, bp)
    assert np.max(np.abs(ilu.export()[2] - G["ilu_val"])) <= 1e-12 * np.abs(G["ilu_val"]).max()
    f1rp, f1ci, f1v = orc.ILU(rp, ci, val, 1, bp).export()
    assert np.array_equal(f1rp, G["ilu1_rowptr"]) and np.array_equal(f1ci, G["ilu1_colidx"])
    assert np.max(np.abs(f1v - G["ilu1_val"])) <= 1e-12 * np.abs(G["ilu1_val"]).max()
    x, info, _ = orc.solve(rp, ci, val, b, singular=False, prec="ilu", ilu=ilu)
    assert info.iters == int(G["iters"][0]) and np.linalg.norm(x - G["x"]) <= 1e-9 * np.linalg.norm(G["x"])
    amg = orc.AMG(rp, ci, val, theta=0.05, block=64, coarse_max=16)
    assert np.array_equal(amg.aggregates(0), G["amg_aggregates"])


def test_schwarz_oracle_against_independent_construction():
    """oracle/isph_schwarz_oracle.c (Ifpack_AdditiveSchwarz<ILU>, precond_ifpack.h:28-75) against a construction
    with scipy index arithmetic + the one-block ILU: subdomain rows, apply (Add / Zero), and the special cases
    overlap 0 == block-Jacobi ILU, one subdomain == whole-matrix ILU(k) whatever the overlap."""
    pr = Problem(tgv_spec(dim=3, n=12, mode=workload.ADVECT, brick=4))
    rp, ci, val, b = pr.poisson()
    N = pr.n
    A = sps.csr_matrix((val, ci, rp), shape=(N, N))
    block = 432
    own = np.arange(0, N + block, block).clip(0, N).astype(np.int32)
    r = np.random.default_rng(1).standard_normal(N)
    for overlap, comb in ((1, "add"), (1, "zero"), (2, "zero")):
        S = orc.Schwarz(rp, ci, val, 0, own, overlap, comb)
        rows, lp = S.export()[:2]
        zz = np.zeros(N)
        for s in range(len(own) - 1):
            o = np.arange(own[s], own[s + 1])
            rws, cur, layer = o, set(o.tolist()), o
            for _ in range(overlap):
                ext = np.array(sorted(set(np.unique(A[layer].indices).tolist()) - cur), dtype=np.int64)
                rws = np.concatenate([rws, ext]); cur |= set(ext.tolist()); layer = ext
            assert np.array_equal(rows[lp[s]:lp[s + 1]], rws)
            Sm = A[rws][:, rws].tocsr(); Sm.sort_indices()
            zl = orc.ILU(Sm.indptr.astype(np.int32), Sm.indices.astype(np.int32), Sm.data.copy(), 0).apply(
                np.ascontiguousarray(r[rws]))
            if comb == "add":
                zz[rws] += zl
            else:
                zz[o] += zl[:len(o)]
        assert np.abs(S.apply(r) - zz).max() <= 1e-14 * np.abs(zz).max()
    assert np.array_equal(orc.Schwarz(rp, ci, val, 1, own, 0).apply(r), orc.ILU(rp, ci, val, 1, own).apply(r))
    assert np.array_equal(orc.Schwarz(rp, ci, val, 1, None, 3).apply(r), orc.ILU(rp, ci, val, 1).apply(r))
    # restricted overlap-1 Schwarz needs fewer iterations than block-Jacobi on the same blocks
    it = {}
    for name, S in (("bj", orc.Schwarz(rp, ci, val, 0, own, 0)), ("ras", orc.Schwarz(rp, ci, val, 0, own, 1, "zero"))):
        x, info, _ = orc.solve(rp, ci, val, b, singular=True, prec="schwarz", schwarz=S)
        assert info.converged
        it[name] = info.iters
    assert it["ras"] < it["bj"]


def test_amg_ml_uncoupled_aggregation_and_whole_level_sgs_gap():
    """AMG fidelity: the oracle restates ML's own sequential Uncoupled aggregation (phases 1-3) and its
    processor-wide Gauss-Seidel next to the device variant (MIS-2 roots, block-local sweeps).  Both are valid
    aggregations (partition of the connected nodes, P reproduces the null vector); the measured iteration gap on the
    3-D TGV system: aggregation algorithm <= 2 iterations, smoother locality <= 50 % (scripts/amg_fidelity.py:
    19 / 15 / 18 / 14 iterations at 32^3 for mis2+block / mis2+whole / ml+block / ml+whole)."""
    pr = Problem(tgv_spec(dim=3, n=20, mode=workload.ADVECT, brick=4))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    nv = np.full(n, 1.0 / np.sqrt(n))
    its = {}
    for agg, whole in (("mis2", False), ("mis2", True), ("ml", False), ("ml", True)):
        G = orc.AMG(rp, ci, val, nullvec=nv, coarse_max=64, aggregation=agg, whole_sgs=whole, block=256)
        assert G.levels >= 2
        a = G.aggregates(0)
        assert a.min() >= 0 and len(np.unique(a)) == G.level_info(1)["rows"]          # partition, every aggregate used
        prp, pci, pv = G.export(0, "P")
        P = sps.csr_matrix((pv, pci, prp), shape=(n, G.level_info(1)["rows"]))
        A = sps.csr_matrix((val, ci, rp), shape=(n, n))
        # smoothed prolongator keeps the (near) null vector in its range: A P 1_c ~ 0 on a zero-row-sum matrix
        nc = np.sqrt(np.bincount(a, weights=nv * nv))
        assert np.abs(A @ (P @ nc)).max() < 1e-10 * np.abs(val).max()
        x, info, _ = orc.solve(rp, ci, val, b, singular=True, prec="amg", amg=G)
        assert info.converged
        its[(agg, whole)] = info.iters
    assert abs(its[("mis2", False)] - its[("ml", False)]) <= 2 and abs(its[("mis2", True)] - its[("ml", True)]) <= 2
    assert its[("ml", True)] <= its[("mis2", False)] <= 1.5 * its[("ml", True)] + 1


# ------------------------------------------------------------------ second pin: the reference's Poisson-Boltzmann table
@pytest.mark.parametrize("N", [16, 32, 64, 128])
def test_pb_harmonic_known_answer_table_pinned(N):
    """sph-script/conv-poisson-boltzmann-harmonic-2d-rev390.txt (fix isph/error, fix_isph_error.cpp:188-345): the
    numbers the reference printed for the manufactured Poisson-Boltzmann problem depend on nothing but the kernel, the
    volumes V_i, the correction tensors G_i / L_i, the corrected (Symmetric-family) Laplacian and the corrected gradient
    (oracle/pb_harmonic.py).  The oracle reproduces total volume to 14 digits and both error norms to >= 10."""
    import pb_harmonic
    ref = pb_harmonic.known_answers()[N]
    r = pb_harmonic.run(N)
    assert ref["particles"] == N * N
    assert abs(r["volume"] - ref["volume"]) <= 1e-13 * ref["volume"]
    assert abs(r["sol_psi"] - ref["sol_psi"]) <= 1e-13 and abs(r["sol_grad"] - ref["sol_grad"]) <= 1e-13
    assert abs(r["err_psi"] - ref["err_psi"]) <= 1e-10 * ref["err_psi"]
    assert abs(r["err_grad"] - ref["err_grad"]) <= 1e-10 * ref["err_grad"]
    assert r["operator_vs_matrix"] <= 1e-11          # functor_laplacian.h's operator form == the assembled rows


@pytest.mark.parametrize("boundary", ["MorrisHolmes", "ConstExtension"])
@pytest.mark.parametrize("N", [32, 64, 128])
def test_pb_channel_known_answer_table_pinned(N, boundary):
    """sph-script/conv-channel-edl-potential-2d-morrisholmes-rev722.txt, both sections (oracle/pb_channel.py): the
    double layer between two charged walls.  Pins the MorrisHolmes mirror coefficient, the particle number density it
    measures distances with, and the solid (Dirichlet) columns of the corrected Laplacian rows to the reference's
    printed digits; "ConstExtension" is the same problem without the mirror."""
    import pb_channel
    ref = pb_channel.known_answers(boundary)[N]
    r = pb_channel.run(N, boundary)
    assert r["particles"] == ref["particles"]
    assert abs(r["volume"] - ref["volume"]) <= 2e-13 * ref["volume"]
    assert abs(r["sol_psi"] - ref["sol_psi"]) <= 1e-14
    assert abs(r["err_psi"] - ref["err_psi"]) <= 1e-10 * ref["err_psi"]
    assert 0.5 < r["xi_min"] < 1.0                   # near-wall fluid: between half and all of the support is fluid


@pytest.mark.parametrize("N", [32, 64])
def test_pb_channel_earlier_table_with_h_102(N):
    """sph-script/conv-channel-edl-potential-2d-morrisholmes-rev406.txt ("MorrisHolmes with h = 1.02 dx"): the same
    channel in another h/dx regime.  Particle count exact, total volume to the table's digits, sol.psi to 1e-14;
    err.psi.norm2 to 4 digits only (that revision's solve is not the one rev722 records; oracle/pb_channel.py says
    which rows are used and why)."""
    import pb_channel
    ref = pb_channel.known_answers("rev406")[N]
    r = pb_channel.run(N, "MorrisHolmes", h_over_dx=1.02)
    assert r["particles"] == ref["particles"]
    assert abs(r["volume"] - ref["volume"]) <= 5e-14 * ref["volume"]
    assert abs(r["sol_psi"] - ref["sol_psi"]) <= 1e-14
    assert abs(r["err_psi"] - ref["err_psi"]) <= 5e-4 * ref["err_psi"]


def test_amg_with_null_vector_masked_to_the_fluid_rows():
    """cavity Poisson system (wall Neumann rows, null vector = fluid mask): aggregates of wall particles have an empty
    coarse row; the smoother must leave them at zero.  Converges, and in fewer iterations than block ILU(0)."""
    from isph_amd import workload
    p = workload.make_cavity(12, wall=4)
    n, nall = p["nlocal"], p["nall"]
    colmap = workload.single_rank_colmap(p)
    P = orc.Particles(p, colmap, kinds=p["kinds"])
    P.precompute(corrections=True)
    rng = np.random.default_rng(0)
    vstar = np.zeros((nall, 3))
    vstar[:n] = rng.standard_normal((n, 3)) * (p["type"][:n, None] == 1)
    vstar = np.ascontiguousarray(vstar[colmap])
    rp, ci, val, b = P.poisson(p["dt"], p["rho"], vstar, antisym=True, singular=orc.NULLSPACE, normal=p["normal"])
    mask = (p["type"][:n] == 1).astype(np.int32)
    G = orc.AMG(rp, ci, val, nullvec=mask / np.sqrt(float(mask.sum())), block=512)
    assert np.all(np.isfinite(G.apply(b)))
    x, ia, _ = orc.solve(rp, ci, val, b, singular=True, null_mask=mask, prec="amg", amg=G)
    bp = np.arange(0, n + 512, 512).clip(0, n).astype(np.int32)
    _, ii, _ = orc.solve(rp, ci, val, b, singular=True, null_mask=mask, prec="ilu", ilu=orc.ILU(rp, ci, val, 0, bp))
    assert ia.converged == 1 and ii.converged == 1 and ia.iters < ii.iters


def test_oracle_kernels_equal_the_reference_kernel_classes():
    """oracle/_ref/libisph_refkernels.so = the reference's OWN KernelFuncWendland / Quintic / Cubic (kernel*.h, the only
    sources of the path that build without Trilinos and LAMMPS), compiled where they lie by oracle/build.py.  The oracle's
    W and dW/dr must be those functions: same value to the last bit or two over the whole support, both dimensions,
    including the breakpoints and the cut radius."""
    import ctypes
    import importlib.util
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_oracle_build", os.path.join(ROOT, "oracle", "build.py"))
    ob = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ob)
    path = ob.build_ref()
    if not (path and os.path.exists(path)):
        pytest.skip("no /root/reference here and no prebuilt oracle/_ref")
    ref = ctypes.CDLL(path)
    ref.ref_kernel_table.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]
    # the particle-kind filter of every functor (filter.h:33-57) over all kinds of pair_isph.h:113-138 and all filters in use
    kinds = [99, 12, 127, 1, 2, 4, 8, 32, 64]
    for fi in (99, 12, 127):
        for ik in kinds:
            assert ref.ref_filter_yes1(fi, ik) == orc.lib().orc_filter_yes1(fi, ik)
            for fj in (99, 12, 127):
                for jk in kinds:
                    assert ref.ref_filter_yes2(fi, fj, ik, jk) == orc.lib().orc_filter_yes2(fi, fj, ik, jk)
    # FilterMatchBinary (filter.h:83-104) of the solute-transport / applied-potential functors: (Fluid, Fluid - BufferNeumann), (Fluid, Fluid)
    for fi in (99, 12):
        for ik in kinds:
            assert ref.ref_filter_match_yes1(fi, ik) == orc.lib().orc_filter_yes1(fi | orc.FILTER_MATCH, ik)
            for fj in (99, 99 - 64, 127):
                for jk in kinds:
                    assert ref.ref_filter_match_yes2(fi, fj, ik, jk) == orc.lib().orc_filter_yes2(fi | orc.FILTER_MATCH, fj, ik, jk)
    # the pair operator of every gradient / divergence / Laplacian functor (functor.h:9-20) and MirrorNothing (mirror.h:19)
    ref.ref_sph_operator.restype = ctypes.c_double
    ref.ref_sph_operator.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double]
    ref.ref_mirror_nothing.restype = ctypes.c_double
    ref.ref_mirror_nothing.argtypes = [ctypes.c_double]
    orc.lib().orc_sph_operator.restype = ctypes.c_double
    orc.lib().orc_sph_operator.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double]
    for fi, fj in ((1.5, -0.25), (0.0, 3.0), (-2.0, -2.0), (1e-300, 1e300)):
        for a in (0, 1):
            assert ref.ref_sph_operator(a, fi, fj) == orc.lib().orc_sph_operator(a, fi, fj)
    assert ref.ref_mirror_nothing(0.3) == 1.0
    rng = np.random.default_rng(1)
    for kernel, name, support in ((0, "wendland", 2.0), (1, "quintic", 3.0), (2, "cubic", 2.0)):
        for dim in (2, 3):
            for h in (1.5 * 2 * np.pi / 100, 0.037, 1.0):
                s = np.concatenate([rng.uniform(0.0, support + 0.3, 400), np.arange(0.0, support + 0.51, 0.5),
                                    np.nextafter(np.arange(1.0, support + 0.1, 1.0), 0.0)])
                r = np.ascontiguousarray(s * h)
                w, dw = np.zeros_like(r), np.zeros_like(r)
                ref.ref_kernel_table(kernel, dim, h, len(r), r.ctypes.data, w.ctypes.data, dw.ctypes.data)
                wo = np.array([orc.kernel_val(name, dim, float(x), h) for x in r])
                dwo = np.array([orc.kernel_dval(name, dim, float(x), h) for x in r])
                scale_w, scale_d = np.abs(w).max(), np.abs(dw).max()
                assert np.max(np.abs(wo - w)) <= 4e-16 * scale_w, (name, dim, h)
                assert np.max(np.abs(dwo - dw)) <= 4e-16 * scale_d, (name, dim, h)
                out = np.abs(r / h) >= support                     # the ratio the kernels themselves form
                assert np.all(w[out] == 0.0) and np.all(wo[out] == 0.0) and out.sum() > 5


def test_amg_gauss_seidel_efficient_symmetric_cycle_is_what_it_says():
    """The oracle's restatement of ml.xml's smoother ("ML Gauss-Seidel" with "smoother: Gauss-Seidel efficient symmetric",
    bench-script/hopper/tgv/1728/ml.xml) against an independent two-level V cycle written with SciPy: `sweeps` block-local
    FORWARD sweeps x += (D+L_B)^-1 (b - A x), the coarse correction through the exported P and coarse operator, `sweeps`
    block-local BACKWARD sweeps x += (D+U_B)^-1 (b - A x)."""
    import scipy.sparse.linalg as spla
    from problems import wall_types
    pr = Problem(tgv_spec(dim=2, n=32, mode=workload.JITTER, brick=8), singular=orc.NOT_SINGULAR,
                 kinds=[orc.FLUID, orc.SOLID], types=wall_types)
    rp, ci, val, b = pr.poisson()
    n = pr.n
    block, sweeps = 256, 3
    G = orc.AMG(rp, ci, val, nullvec=None, coarse_max=256, block=block, sweeps=sweeps, smoother=1)
    assert G.levels == 2
    A = sps.csr_matrix((val, ci, rp), shape=(n, n))
    prp, pci, pv = G.export(0, "P")
    r1, c1, v1 = G.export(1, "A")
    nc = G.level_info(1)["rows"]
    P = sps.csr_matrix((pv, pci, prp), shape=(n, nc))
    Ac = sps.csr_matrix((v1, c1, r1), shape=(nc, nc))
    # block-diagonal lower / upper triangles (diagonal included)
    rows, cols = A.nonzero()
    same = (rows // block) == (cols // block)
    Ab = sps.csr_matrix((np.asarray(A[rows[same], cols[same]]).ravel(), (rows[same], cols[same])), shape=(n, n))
    Lb, Ub = sps.tril(Ab, format="csr"), sps.triu(Ab, format="csr")

    def cycle(r):
        x = np.zeros(n)
        for _ in range(sweeps):
            x = x + spla.spsolve_triangular(Lb, r - A @ x, lower=True)
        xc = spla.spsolve(Ac.tocsc(), P.T @ (r - A @ x))
        x = x + P @ xc
        for _ in range(sweeps):
            x = x + spla.spsolve_triangular(Ub, r - A @ x, lower=False)
        return x

    r = np.random.default_rng(11).standard_normal(n)
    zo, zs = G.apply(r), cycle(r)
    assert np.linalg.norm(zo - zs) <= 1e-11 * np.linalg.norm(zs)
    # and it is a different cycle from the symmetric-sweep one
    Gs = orc.AMG(rp, ci, val, nullvec=None, coarse_max=256, block=block, sweeps=sweeps, smoother=0)
    assert np.linalg.norm(Gs.apply(r) - zo) > 1e-3 * np.linalg.norm(zo)
