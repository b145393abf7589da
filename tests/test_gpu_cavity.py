"""-m gpu: BASELINE configs[3] -- 3-D lid-driven cavity (sph-script/lid-driven-cavity-3d.{m,lmp}, lid-driven-cavity.xml):
closed box, solid walls with normals, moving lid; block Helmholtz (functor_incomp_navier_stokes_block_helmholtz.h) ->
solveBlockProblem, then the pressure Poisson system with wall Neumann rows and the null-space mask.
Oracle parity at 16^3 sites, size-independent properties at the configuration's 2 M particles."""
import numpy as np
import pytest
import scipy.sparse as sps

from isph_amd import hip, workload
import oracle as orc

from sampled_check import assert_spmv_matches_host_on_sampled_rows

pytestmark = pytest.mark.gpu

THETA, BETA = 1.0, 0.1          # lid-driven-cavity.xml: theta 1.0, beta 0.1


def _oracle_particles(p, colmap):
    P = orc.Particles(p, colmap, kinds=p["kinds"])
    P.precompute(corrections=True)
    return P


@pytest.mark.parametrize("antisym", [True, False])
def test_cavity_small_matches_oracle(gpu_ctx, antisym):
    p = workload.make_cavity(8, wall=4, brick=(4, 4, 4), lid_inset=1, jitter=0.02)
    assert (p["type"][:p["nlocal"]] == 3).sum() > 0
    colmap = workload.single_rank_colmap(p)
    n, nall, dim = p["nlocal"], p["nall"], 3
    P = _oracle_particles(p, colmap)
    zeros3, g = np.zeros((nall, 3)), np.zeros(3)
    pres = np.zeros(nall)
    # ---- block Helmholtz: assembly block by block, then the block solve
    rp, ci, vals, b = P.block_helmholtz(p["dt"], THETA, BETA, p["nu"], p["rho"], pres, zeros3, g, p["v"],
                                        normal=p["normal"], antisym=antisym)
    blocks, bg = hip.assemble_block_helmholtz(gpu_ctx, p, colmap, p["dt"], THETA, BETA, p["nu"], p["rho"], pres, zeros3,
                                              g, np.ascontiguousarray(p["v"]), normal=p["normal"], antisym=antisym,
                                              vfrac=P.vfrac, Gc=P.Gc, Lc=None if antisym else P.Lc, kinds=p["kinds"])
    scale = np.abs(vals).max()
    for ib in range(dim):
        for jb in range(dim):
            rg, cg, vg = blocks[ib][jb].export_csr()
            assert np.array_equal(rg, rp) and np.array_equal(cg, ci)
            assert np.max(np.abs(vg - vals[ib * dim + jb])) <= 1e-12 * scale
    assert np.max(np.abs(bg - b.ravel())) <= 1e-12 * max(np.abs(b).max(), 1e-300)
    big = sps.bmat([[sps.csr_matrix((vals[ib * dim + jb], ci, rp), shape=(n, n)) for jb in range(dim)]
                    for ib in range(dim)], format="csr")
    big.sort_indices()
    x0 = np.ascontiguousarray(p["v"][:n].T).ravel()                # initial guess v^n (pair_isph.cpp:925-927)
    xo, io = orc.solve_block(big.indptr, big.indices, big.data, b.ravel(), dim, x0=x0, prec="none")
    x = x0.copy()
    info = hip.solve_block(gpu_ctx, blocks, bg.copy(), x)
    assert info.converged == 1 and io.converged == 1 and abs(info.iters - io.iters) <= 1
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
    # ---- Poisson with the wall Neumann rows, null space masked to the fluid rows (pair_isph.cpp:996-1003)
    vstar = np.zeros((nall, 3))
    vstar[:n] = x.reshape(dim, n).T
    vstar = np.ascontiguousarray(vstar[colmap])
    rp2, ci2, val2, b2 = P.poisson(p["dt"], p["rho"], vstar, antisym=antisym, singular=orc.NULLSPACE, normal=p["normal"])
    A, bp = hip.assemble_poisson(gpu_ctx, p, colmap, p["dt"], p["rho"], vstar, antisym=antisym, vfrac=P.vfrac, Gc=P.Gc,
                                 Lc=None if antisym else P.Lc, kinds=p["kinds"], normal=p["normal"])
    rg, cg, vg = A.export_csr()
    assert np.array_equal(cg, ci2) and np.max(np.abs(vg - val2)) <= 1e-12 * np.abs(val2).max()
    assert np.max(np.abs(bp - b2)) <= 1e-12 * np.abs(b2).max()
    mask = (p["type"][:n] == 1).astype(np.int32)
    xp = np.zeros(n)
    ip = hip.solve(gpu_ctx, A, bp.copy(), xp, prec=hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 512), singular=True,
                   null_mask=mask)
    bpt = np.arange(0, n + 512, 512).clip(0, n).astype(np.int32)
    xpo, ipo, _ = orc.solve(rp2, ci2, val2, b2, singular=True, null_mask=mask, prec="ilu",
                            ilu=orc.ILU(rp2, ci2, val2, 0, bpt))
    assert ip.converged == 1 and ipo.converged == 1 and abs(ip.iters - ipo.iters) <= 1
    assert np.linalg.norm(xp - xpo) <= 1e-6 * np.linalg.norm(xpo)


def test_cavity_poisson_sa_amg_with_masked_null_vector(gpu_ctx):
    """lid-driven-cavity.xml:30-32 selects ML.  The singular pressure system hands ML the null vector masked to the fluid
    rows (pair_isph.cpp:996-1003, precond_ml.h:97-127): aggregates made of wall particles only carry none of it, their
    column of P and their row of R A P are empty.  Those coarse unknowns stay at zero (no division by the zero pivot) --
    before that guard the cycle produced NaN and FGMRES ran into its iteration limit.  Device vs oracle: same hierarchy,
    iterations +-1, x <= 1e-6; and the preconditioner beats block ILU(0) on this system."""
    p = workload.make_cavity(14, wall=4, brick=(4, 4, 4), jitter=0.02)
    colmap = workload.single_rank_colmap(p)
    n, nall = p["nlocal"], p["nall"]
    P = _oracle_particles(p, colmap)
    rng = np.random.default_rng(5)
    vstar = np.zeros((nall, 3))
    vstar[:n] = rng.standard_normal((n, 3)) * (p["type"][:n, None] == 1)
    vstar = np.ascontiguousarray(vstar[colmap])
    rp, ci, val, b = P.poisson(p["dt"], p["rho"], vstar, antisym=True, singular=orc.NULLSPACE, normal=p["normal"])
    mask = (p["type"][:n] == 1).astype(np.int32)
    nv = mask / np.sqrt(float(mask.sum()))
    G = orc.AMG(rp, ci, val, nullvec=nv, block=512)
    xo, io, _ = orc.solve(rp, ci, val, b, singular=True, null_mask=mask, prec="amg", amg=G)
    bp = np.arange(0, n + 512, 512).clip(0, n).astype(np.int32)
    _, ii, _ = orc.solve(rp, ci, val, b, singular=True, null_mask=mask, prec="ilu", ilu=orc.ILU(rp, ci, val, 0, bp))
    assert io.converged == 1 and ii.converged == 1 and io.iters < ii.iters
    A, bg = hip.assemble_poisson(gpu_ctx, p, colmap, p["dt"], p["rho"], vstar, antisym=True, vfrac=P.vfrac, Gc=P.Gc,
                                 kinds=p["kinds"], normal=p["normal"])
    M = hip.PrecondAMG(gpu_ctx, A, nullvec=nv, params=hip.AmgParams(block=512))
    assert M.levels == G.levels and M.levels >= 2
    x = np.zeros(n)
    info = hip.solve(gpu_ctx, A, bg.copy(), x, prec=M, singular=True, null_mask=mask)
    assert info.converged == 1 and abs(info.iters - io.iters) <= 1
    assert np.all(np.isfinite(x)) and np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)


@pytest.mark.parametrize("prec", ["bjacobi-ilu0", "sa-amg"])
@pytest.mark.parametrize("morris", [False, True])
@pytest.mark.parametrize("singular", ["NullSpace", "PinZero", "DoubleDiag", "NotSingular"])
def test_wall_bounded_poisson_every_singular_mode_and_preconditioner(gpu_ctx, singular, morris, prec):
    """The combinations the reference's scripts select for wall-bounded flows (sph-script/*.xml: "Singular Poisson" in
    {NullSpace, PinZero, NotSingular}, boundary MorrisHolmes or none, Ifpack or ML) on a 36^3 closed box without wall
    normals (solid rows are identity rows): every one must converge to 1e-8 with a finite solution, and the residual
    re-computed with an independent SpMV must agree."""
    p = workload.make_cavity(28, wall=4)
    colmap = workload.single_rank_colmap(p)
    n, nall = p["nlocal"], p["nall"]
    mode = {"NullSpace": hip.NULLSPACE, "PinZero": hip.PINZERO, "DoubleDiag": hip.DOUBLEDIAG, "NotSingular": hip.NOT_SINGULAR}[singular]
    vf = hip.compute_volumes(gpu_ctx, p, colmap)
    vfrac = np.ascontiguousarray(vf[colmap])
    pnd = np.ascontiguousarray(hip.compute_pnd(gpu_ctx, p, colmap, kinds=p["kinds"])[colmap]) if morris else None
    rng = np.random.default_rng(11)
    vstar = np.zeros((nall, 3))
    vstar[:n] = 0.1 * rng.standard_normal((n, 3)) * (p["type"][:n, None] == 1)
    vstar = np.ascontiguousarray(vstar[colmap])
    A, b = hip.assemble_poisson(gpu_ctx, p, colmap, p["dt"], p["rho"], vstar, antisym=True, singular=mode, vfrac=vfrac,
                                kinds=p["kinds"], pnd=pnd)
    null = singular == "NullSpace"
    mask = (p["type"][:n] == 1).astype(np.int32) if null else None
    if prec == "sa-amg":
        nv = mask / np.sqrt(float(mask.sum())) if null else None
        M = hip.PrecondAMG(gpu_ctx, A, nullvec=nv, params=hip.AmgParams(block=512))
    else:
        M = hip.Precond(gpu_ctx, A, prec, 512)
    x = np.zeros(n)
    bw = b.copy()
    info = hip.solve(gpu_ctx, A, bw, x, prec=M, singular=null, null_mask=mask)
    assert info.converged == 1 and np.all(np.isfinite(x)), (singular, morris, prec, info.iters)
    ax = A.spmv(x)
    assert_spmv_matches_host_on_sampled_rows(A, x, ax, nsamples=16)   # the product the residual rests on, re-done on the host
    r = bw - ax
    if null:                                             # solved in the complement of the masked null vector
        nvv = mask / np.sqrt(float(mask.sum()))
        r -= (r @ nvv) * nvv
    assert np.linalg.norm(r) <= 5e-8 * np.linalg.norm(bw)
    solid = p["type"][:n] >= 2
    assert np.max(np.abs(x[solid])) <= 1e-12 * max(np.max(np.abs(x)), 1e-300)   # identity rows with b = 0


def test_cavity_config3_full_size_properties(gpu_ctx):
    """126^3 = 2 000 376 particles (114^3 fluid + 6 wall layers, the .m script's nn = 6): the 3x3 block Helmholtz system
    and the pressure Poisson system of one time step, assembled and solved on the device with torch-resident arrays.
    Properties: all nine blocks exist on the scalar pattern, block residual <= 2e-8 re-computed
    block by block with independent SpMV calls, the lid drags the fluid (+x velocity under the lid), solid rows keep
    their velocity; Poisson: converged, residual <= 2e-8, pressure orthogonal to the masked null vector."""
    import torch
    dev = torch.device("cuda", 0)
    p = workload.make_cavity(114, wall=6)
    n, nall, dim = p["nlocal"], p["nall"], 3
    assert n == 126 ** 3
    colmap_h = workload.single_rank_colmap(p)
    dp = dict(p)
    for k in ("x", "type", "neigh_ptr", "neigh_idx"):
        dp[k] = torch.from_numpy(np.ascontiguousarray(p[k])).to(dev)
    colmap = torch.from_numpy(colmap_h).to(dev)
    own = torch.from_numpy(p["owner_index"].astype(np.int64)).to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rho, nu, vel, nrm = t(p["rho"]), t(p["nu"]), t(p["v"]), t(p["normal"])
    vf = hip.compute_volumes(gpu_ctx, dp, colmap)
    vfrac = vf[own].contiguous()
    G, _ = hip.compute_corrections(gpu_ctx, dp, colmap, vfrac)        # computePre: G_i for the wall terms
    Gc = G[own].contiguous()
    pres = torch.zeros(nall, dtype=torch.float64, device=dev)
    force = torch.zeros((nall, 3), dtype=torch.float64, device=dev)
    blocks, b = hip.assemble_block_helmholtz(gpu_ctx, dp, colmap, p["dt"], THETA, BETA, nu, rho, pres, force, np.zeros(3),
                                             vel, normal=nrm, vfrac=vfrac, Gc=Gc, kinds=p["kinds"])
    assert all(blocks[i][j] is not None for i in range(3) for j in range(3))
    x = vel[:n].t().contiguous().reshape(-1).clone()
    M = hip.Precond(gpu_ctx, blocks[0][0], "bjacobi-ilu0", 512)
    bw = b.clone()
    info = hip.solve_block(gpu_ctx, blocks, bw, x, prec=M)
    assert info.converged == 1
    res2, bn2 = 0.0, 0.0
    for i in range(3):
        r = b[i * n:(i + 1) * n].clone()
        for j in range(3):
            xj = x[j * n:(j + 1) * n].contiguous()
            y = blocks[i][j].spmv(xj)
            if i == j or (i, j) == (0, 1):   # independent host product on sampled rows of the diagonal blocks and one coupling block
                assert_spmv_matches_host_on_sampled_rows(blocks[i][j], xj, y, nsamples=32)
            r -= y
        res2 += float((r * r).sum())
        bn2 += float((b[i * n:(i + 1) * n] ** 2).sum())
    assert np.sqrt(res2 / bn2) < 2e-8
    vs = x.reshape(3, n).t().contiguous()
    typ = dp["type"][:n]
    assert float((vs[typ == 3][:, 0] - 5.0).abs().max()) < 1e-8        # the lid keeps its velocity (solid:fixed rows)
    assert float(vs[typ == 2].abs().max()) < 1e-8
    idx_y = torch.from_numpy(((p["tag"][:n].astype(np.int64) - 1) // 126) % 126).to(dev)
    under_lid = (typ == 1) & (idx_y == 6 + 114 - 1)
    assert float(vs[under_lid][:, 0].mean()) > 0.0                      # dragged along +x
    # ---- Poisson
    vstar = torch.zeros((nall, 3), dtype=torch.float64, device=dev)
    vstar[:n] = vs
    vstar = vstar[own].contiguous()
    A, bp = hip.assemble_poisson(gpu_ctx, dp, colmap, p["dt"], rho, vstar, vfrac=vfrac, Gc=Gc, kinds=p["kinds"], normal=nrm)
    mask = (p["type"][:n] == 1).astype(np.int32)
    xp = torch.zeros(n, dtype=torch.float64, device=dev)
    bpw = bp.clone()
    ip = hip.solve(gpu_ctx, A, bpw, xp, prec=hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 512), singular=True, null_mask=mask)
    assert ip.converged == 1
    m = torch.from_numpy(mask.astype(np.float64)).to(dev)
    nvec = m / m.norm()
    # The solver solves P A x = P b (P = I - n n^T, PoissonProjection) and then, like the reference
    # (solver_lin_belos.h:215-219), returns x - (x.n) n.  With wall Neumann rows n = mask/|mask| is not an exact null
    # vector of A (those rows couple to fluid columns), so the returned x differs from the Krylov solution by a
    # multiple of n and its projected residual lies along q = P A n: remove that one direction, the rest is <= 2e-8.
    axp = A.spmv(xp)
    assert_spmv_matches_host_on_sampled_rows(A, xp, axp)
    r = bpw - axp
    r -= (r @ nvec) * nvec
    an = A.spmv(nvec)
    an -= (an @ nvec) * nvec
    if float(an.norm()) > 0.0:
        q = an / an.norm()
        r -= (r @ q) * q
    assert float(r.norm() / bpw.norm()) < 2e-8
    assert abs(float(xp @ nvec)) < 1e-10 * float(xp.abs().max())
