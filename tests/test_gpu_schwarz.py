"""-m gpu: Ifpack_AdditiveSchwarz<ILU(k)> with the reference's own semantics (csrc/schwarz.hpp) against the oracle
(oracle/isph_schwarz_oracle.c): one subdomain = the whole matrix ("ilu<k>", what the reference factors on one MPI rank,
precond_ifpack.h:60-74), subdomains beyond the block stream's 1024 rows, "Overlap Level" 1/2 with combine Add / Zero."""
import numpy as np
import pytest

from isph_amd import hip, workload
import oracle as orc
from problems import Problem, tgv_spec

pytestmark = pytest.mark.gpu


def _factor_close(gv, fv):
    return np.max(np.abs(gv - fv) / np.maximum(np.abs(fv), 1e-300 + 1e-10 * np.abs(fv).max())) < 1e-10


@pytest.mark.parametrize("case,fill", [
    (dict(dim=2, n=16, mode=workload.JITTER, brick=8), 0),
    (dict(dim=2, n=33, mode=workload.ADVECT, brick=8), 1),
    (dict(dim=3, n=12, mode=workload.JITTER, brick=4), 1),
    (dict(dim=3, n=16, mode=workload.ADVECT, brick=8), 0),
    (dict(dim=3, n=12, mode=workload.ADVECT, brick=4), 2),
    # Quintic cut 3h: 400+ entries per row, level-1 rows beyond the device pattern kernel's table: the host builds it
    (dict(dim=3, n=12, mode=workload.ADVECT, brick=4, kernel="quintic", cut_over_h=3.0), 1),
])
def test_whole_matrix_iluk_matches_oracle(gpu_ctx, case, fill):
    """"ilu<k>": pattern exact, factor 1e-10, apply 1e-11, GMRES iterations +-1 and x 1e-6 against the oracle's
    one-block ILU(k) (the reference's one-rank configuration; default fill 1)."""
    pr = Problem(tgv_spec(**case))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    ref = orc.ILU(rp, ci, val, fill)
    frp, fci, fv = ref.export()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    M = hip.Precond(gpu_ctx, A, "ilu%d" % fill, 0)
    Ms = hip.PrecondSchwarz.__new__(hip.PrecondSchwarz)
    Ms.ctx, Ms.h, Ms.n = gpu_ctx, M.h, n
    rows, lp, grp, gci, gv = Ms.export()
    Ms.h = None
    assert np.array_equal(rows, np.arange(n)) and list(lp) == [0, n]
    assert np.array_equal(grp, frp) and np.array_equal(gci, fci)
    assert _factor_close(gv, fv)
    r = np.random.default_rng(5).standard_normal(n)
    z, zo = M.apply(r), ref.apply(r)
    assert np.linalg.norm(z - zo) / np.linalg.norm(zo) < 1e-11
    x = np.zeros(n)
    info = hip.solve(gpu_ctx, A, b.copy(), x, prec=M, singular=True)
    xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="ilu", ilu=ref)
    assert info.converged == 1 and io.converged == 1 and abs(info.iters - io.iters) <= 1
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-6


@pytest.mark.parametrize("block,fill,overlap,combine", [
    (512, 0, 1, "add"), (512, 0, 1, "zero"), (512, 1, 1, "add"), (2048, 0, 0, "add"), (2048, 1, 1, "zero"),
    (1000, 0, 2, "zero"), (4096, 0, 1, "add"),
])
def test_schwarz_overlap_matches_oracle(gpu_ctx, block, fill, overlap, combine):
    """subdomains of `block` consecutive rows (beyond the 1024-row limit of the block stream) extended by `overlap`
    layers, combine Add (reference default) / Zero: extended row lists and factor pattern exact, factor 1e-10,
    apply 1e-11, iterations +-1, x 1e-6."""
    pr = Problem(tgv_spec(dim=3, n=16, mode=workload.ADVECT))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    own = np.arange(0, n + block, block).clip(0, n).astype(np.int32)
    ref = orc.Schwarz(rp, ci, val, fill, own, overlap, combine)
    orow, olp, orp, oci, ov = ref.export()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    M = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=fill, overlap=overlap, combine=combine, block_size=block)
    rows, lp, grp, gci, gv = M.export()
    assert np.array_equal(rows, orow) and np.array_equal(lp, olp)
    assert np.array_equal(grp, orp) and np.array_equal(gci, oci)
    assert _factor_close(gv, ov)
    r = np.random.default_rng(7).standard_normal(n)
    z, zo = M.apply(r), ref.apply(r)
    assert np.linalg.norm(z - zo) / np.linalg.norm(zo) < 1e-11
    x = np.zeros(n)
    info = hip.solve(gpu_ctx, A, b.copy(), x, prec=M, singular=True)
    xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="schwarz", schwarz=ref)
    assert info.converged == 1 and io.converged == 1 and abs(info.iters - io.iters) <= 1
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-6


def test_config0_2d_tgv_cg_whole_matrix_ilu0(gpu_ctx):
    """BASELINE configs[0] as the reference runs it on one rank: 2-D TGV 128^2 = 16 384 particles, Wendland,
    Block CG + ILU(0) of the WHOLE matrix (no block decomposition), tol 1e-6: same iteration count as the oracle
    (+-1), same pressure vector (1e-6)."""
    pr = Problem(tgv_spec(dim=2, n=128, mode=workload.ADVECT))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    assert n == 16384
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    M = hip.Precond(gpu_ctx, A, "ilu0", 0)
    x = np.zeros(n)
    info = hip.solve(gpu_ctx, A, b.copy(), x, prec=M, singular=True, params=hip.SolverParams(solver_type=1, tol=1e-6))
    xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="ilu", ilu=orc.ILU(rp, ci, val, 0),
                          params=orc.SolverParams(solver_type=1, tol=1e-6))
    assert info.converged == 1 and io.converged == 1 and abs(info.iters - io.iters) <= 1, (info.iters, io.iters)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-6


@pytest.mark.parametrize("fill,combine", [(0, "add"), (1, "add"), (0, "zero")])
def test_overlap_level_one_across_ranks_through_the_self_peer_plan(fill, combine):
    """isph_prec_create_overlap = Ifpack_AdditiveSchwarz<ILU(k)> with "Overlap Level" 1 on more than one rank
    (precond_ifpack.h:43,60-74).  One GPU: the periodic images are ghost columns received from the rank itself, so the
    whole machinery runs -- row extension (dist.extend_rows; against the global operator in tests/test_dist_cpu.py),
    RCCL gather of the ghost part of r, ILU(k) of the extended matrix on the level-scheduled path, RCCL return of the
    ghost corrections with combine Add.  Reference: the same composition with the oracle's ILU(k) and numpy indexing."""
    from isph_amd import dist
    from problems import Problem, tgv_spec
    pr = Problem(tgv_spec(dim=3, n=12, mode=workload.JITTER))
    rpf, cif, valf, b = pr.poisson()
    ctx = hip.Context(0, rank=0, nranks=1, uid=hip.Context.unique_id())
    try:
        plan = dist.make_self_halo_plan(pr.parts)
        n, next_ = pr.n, plan.ncol
        A, bg = hip.assemble_poisson(ctx, pr.parts, plan.colmap, pr.spec.dt, pr.parts["rho"],
                                     np.ascontiguousarray(pr.parts["v"]), vfrac=pr.P.vfrac, ncol=plan.ncol)
        A.set_halo(plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr)
        rp, ci, val = A.export_csr()
        rpe, cie, ve = dist.extend_rows(plan, rp, ci, val, None)
        assert len(rpe) == next_ + 1 and cie.max() < next_ and np.array_equal(rpe[:n + 1], rp)
        Aext = hip.Matrix.from_csr(ctx, rpe, cie, ve)
        M = hip.PrecondOverlap(ctx, Aext, plan, level_of_fill=fill, combine=combine)
        F = orc.ILU(rpe, cie, ve, fill)                                   # the extended subdomain as one block
        ridx, sidx = plan.recv_idx.astype(np.int64), plan.send_idx.astype(np.int64)

        def reference(r):
            zext = F.apply(np.concatenate([r, r[ridx]]))
            z = zext[:n].copy()
            if combine == "add":
                np.add.at(z, sidx, zext[n:])
            return z

        r = np.random.default_rng(5).standard_normal(n)
        zo = reference(r)
        zg = M.apply(r)
        assert np.max(np.abs(zg - zo)) <= 1e-10 * np.abs(zo).max()
        x = np.zeros(n)
        info = hip.solve(ctx, A, bg.copy(), x, prec=M, singular=True)
        xo, io, _ = orc.solve(rpf, cif, valf, b, singular=True, prec="none")
        assert info.converged == 1 and info.iters < io.iters              # it is a preconditioner
        assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
    finally:
        ctx.close()


@pytest.mark.parametrize("n,fill,block,overlap", [(16, 0, 0, 0), (16, 1, 0, 0), (32, 0, 0, 0), (24, 1, 0, 0), (24, 0, 2048, 1)])
def test_synchronisation_free_sweeps_equal_the_level_launches(gpu_ctx, n, fill, block, overlap):
    """The persistent form of the factorisation and of the two triangular sweeps (one launch each; a row waits for the
    very words it depends on, csrc/schwarz.hpp k_gilu_factor_sf / k_gilu_solve_run) against one launch per dependency
    level, on whole-matrix ILU(0)/ILU(1) with thousands of levels and on overlapping subdomains.  The factor: BIT FOR
    BIT (same operations in the same order).  The application: the sweeps sum a row's products in another order (terms
    from outside the workgroup's run of rows first, then the ones handed on through LDS), so equal to a few units of
    rounding -- and bit for bit from one application to the next, whatever the order the rows happened to finish in.
    Repeated applications reuse the flag words."""
    sp = tgv_spec(dim=3, n=n, mode=workload.JITTER)
    p = workload.make_tgv(sp)
    colmap = workload.single_rank_colmap(p)
    vf = hip.compute_volumes(gpu_ctx, p, colmap)
    A, b = hip.assemble_poisson(gpu_ctx, p, colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]),
                                vfrac=np.ascontiguousarray(vf[p["owner_index"]]))
    N = p["nlocal"]
    Ms = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=fill, overlap=overlap, block_size=block)
    Ml = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=fill, overlap=overlap, block_size=block, level_launches=True)
    info, infol = Ms.schwarz_info(), Ml.schwarz_info()
    assert info.pop("persistent") == 1 and infol.pop("persistent") == 0     # the two forms really are compared
    assert info == infol and info["levels_l"] > (100 if block == 0 else 10)
    for a, c in zip(Ms.export(), Ml.export()):
        assert np.array_equal(a, c)
    rng = np.random.default_rng(n)
    for rep in range(3):
        r = rng.standard_normal(N)
        zs, zl = Ms.apply(r), Ml.apply(r)
        assert np.all(np.isfinite(zs)) and np.max(np.abs(zs - zl)) <= 1e-13 * np.abs(zl).max()
        assert np.array_equal(zs, Ms.apply(r))                  # reproducible
    xs, xl = np.zeros(N), np.zeros(N)
    i1 = hip.solve(gpu_ctx, A, b.copy(), xs, prec=Ms, singular=True)
    i2 = hip.solve(gpu_ctx, A, b.copy(), xl, prec=Ml, singular=True)
    assert i1.converged == 1 and abs(i1.iters - i2.iters) <= 1 and np.linalg.norm(xs - xl) <= 1e-7 * np.linalg.norm(xl)
    xr = np.zeros(N)
    i3 = hip.solve(gpu_ctx, A, b.copy(), xr, prec=Ms, singular=True)
    assert i3.iters == i1.iters and np.array_equal(xs, xr)     # the whole solve is reproducible
    for o in (Ms, Ml, A):
        o.close()


def test_form_of_the_sweeps_follows_the_width_of_the_levels(gpu_ctx):
    """create() picks the persistent launches for long, narrow dependency chains (one subdomain = the whole matrix), and for
    many small subdomains -- where a level holds thousands of short rows (csrc/schwarz.hpp kGiluWideLevel, measured in
    profiles/r03_schwarz_syncfree.txt) -- one workgroup per subdomain (round 4, k_gilu_solve_sub; the launch per level it
    replaces stays selectable and must agree to rounding); all give the oracle's preconditioner."""
    sp = tgv_spec(dim=3, n=64, mode=workload.JITTER)
    p = workload.make_tgv(sp)
    colmap = workload.single_rank_colmap(p)
    vf = hip.compute_volumes(gpu_ctx, p, colmap)
    A, b = hip.assemble_poisson(gpu_ctx, p, colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]),
                                vfrac=np.ascontiguousarray(vf[p["owner_index"]]))
    N = p["nlocal"]
    whole = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=0, overlap=0, block_size=0)
    small = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=0, overlap=0, block_size=64)
    iw, ism = whole.schwarz_info(), small.schwarz_info()
    assert iw["persistent"] == 1 and iw["nloc"] // iw["levels_l"] < 4096
    assert ism["persistent"] == 2 and ism["nloc"] // min(ism["levels_l"], ism["levels_u"]) >= 4096
    forced = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=0, overlap=0, block_size=64, level_launches=True)
    assert forced.schwarz_info()["persistent"] == 0
    r = np.random.default_rng(5).standard_normal(N)
    zs, zf = small.apply(r), forced.apply(r)
    assert np.linalg.norm(zs - zf) <= 1e-13 * np.linalg.norm(zf)
    x = np.zeros(N)
    info = hip.solve(gpu_ctx, A, b.copy(), x, prec=small, singular=True)
    assert info.converged == 1
    for o in (whole, small, forced, A):
        o.close()


@pytest.mark.parametrize("block,fill,overlap,combine", [(256, 0, 1, "add"), (256, 0, 1, "zero"), (192, 1, 1, "add"), (256, 1, 1, "zero"),
                                                       (256, 0, 0, "add"), (200, 0, 1, "add"), (200, 0, 1, "zero"), (200, 0, 0, "zero")])
def test_many_small_subdomains_one_workgroup_per_subdomain(gpu_ctx, block, fill, overlap, combine):
    """>= 32 subdomains of <= 4096 extended rows take the form with ONE launch per application (k_gilu_solve_sub: a
    workgroup per subdomain, its part of the vector in LDS, a barrier per level).  Against the oracle like every other form
    (row lists / pattern exact, factor 1e-10, application 1e-11, iterations +-1, x 1e-6) and against the launch per level on
    the same factor (same bits in the factor, application equal to rounding)."""
    pr = Problem(tgv_spec(dim=3, n=24, mode=workload.ADVECT))
    rp, ci, val, b = pr.poisson()
    n = pr.n
    own = np.arange(0, n + block, block).clip(0, n).astype(np.int32)
    ref = orc.Schwarz(rp, ci, val, fill, own, overlap, combine)
    orow, olp, orp, oci, ov = ref.export()
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    M = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=fill, overlap=overlap, combine=combine, block_size=block)
    assert M.schwarz_info()["persistent"] == 2 and M.schwarz_info()["nsub"] >= 32
    Ml = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=fill, overlap=overlap, combine=combine, block_size=block, level_launches=True)
    assert Ml.schwarz_info()["persistent"] == 0
    rows, lp, grp, gci, gv = M.export()
    assert np.array_equal(rows, orow) and np.array_equal(lp, olp)
    assert np.array_equal(grp, orp) and np.array_equal(gci, oci)
    assert _factor_close(gv, ov)
    assert np.array_equal(gv, Ml.export()[4])
    r = np.random.default_rng(7).standard_normal(n)
    z, zl, zo = M.apply(r), Ml.apply(r), ref.apply(r)
    assert np.linalg.norm(z - zo) / np.linalg.norm(zo) < 1e-11
    assert np.linalg.norm(z - zl) / np.linalg.norm(zl) < 1e-13
    assert np.array_equal(M.apply(r), z)                                  # the same bits from one application to the next
    x = np.zeros(n)
    info = hip.solve(gpu_ctx, A, b.copy(), x, prec=M, singular=True)
    xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="schwarz", schwarz=ref)
    assert info.converged == 1 and io.converged == 1 and abs(info.iters - io.iters) <= 1
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-6


def test_many_subdomains_too_large_for_one_workgroup_take_the_host_path(gpu_ctx):
    """>= 32 subdomains whose extended row lists outgrow the one-workgroup form (4096 rows): the device-side row lists raise
    their flag and the set-up falls back to the host path with a launch per level -- same row lists, pattern and
    application as the oracle."""
    pr = Problem(tgv_spec(dim=3, n=40, mode=workload.ADVECT))
    rp, ci, val, b = pr.poisson()
    n, block = pr.n, 1600
    own = np.arange(0, n + block, block).clip(0, n).astype(np.int32)
    ref = orc.Schwarz(rp, ci, val, 0, own, 1, "zero")
    orow, olp, orp, oci, ov = ref.export()
    assert len(own) - 1 >= 32 and np.diff(olp).max() > 4096
    A = hip.Matrix.from_csr(gpu_ctx, rp, ci, val)
    M = hip.PrecondSchwarz(gpu_ctx, A, level_of_fill=0, overlap=1, combine="zero", block_size=block)
    assert M.schwarz_info()["persistent"] != 2
    rows, lp, grp, gci, gv = M.export()
    assert np.array_equal(rows, orow) and np.array_equal(lp, olp)
    assert np.array_equal(grp, orp) and np.array_equal(gci, oci)
    assert _factor_close(gv, ov)
    r = np.random.default_rng(11).standard_normal(n)
    z, zo = M.apply(r), ref.apply(r)
    assert np.linalg.norm(z - zo) / np.linalg.norm(zo) < 1e-11
    M.close(); A.close()
