"""-m gpu: the multi-rank path with DIFFERENT peers per rank, on one GPU.

Every rank is a thread with its own context over the host-staged transport (tests/ranks.py, tests/cpp/rank_threads.cpp;
include/isph_hip.h isph_ctx_create_hostcomm) -- the code that runs is the library's multi-rank code: forward comm of the
volumes (isph_halo_forward), row-local assembly with ghost columns and the rank-0-only singular modes, the halo exchange
on the second stream overlapped with the interior slices, boundary slices, all-reduced dots, block-Jacobi ILU per rank,
"Overlap Level" 1 with the Add return to a different owner, rank-local SA-AMG.  Partitioning = the reference's: rows =
the rank's particles, columns = owned + ghost tags (pair_isph.cpp:1258-1259; Epetra_Import inside
Epetra_CrsMatrix::Apply, solver_lin.h:133; Norm2/Dot all-reduces, solver_lin.cpp:72-74).

The checker is the SINGLE-RANK oracle on the whole lattice, permuted to the rank-concatenated row order: assembled
rows entry by entry, iteration counts +-1 (the dots are reduced in another order), pressure <= 1e-6."""
import numpy as np
import pytest
import scipy.sparse as sps

from isph_amd import dist, hip, workload
import oracle as orc
from ranks import RankGroup, empty_parts

pytestmark = pytest.mark.gpu

BLOCK = 256


def _spec(dim, pgrid, n, rank=None):
    ncell = tuple(n * g for g in pgrid[:dim])
    kw = dict(dim=dim, ncell=ncell, brick=(4,) * dim, origin=(0.5,) * dim if dim == 2 else (0.0,) * 3, mode=workload.JITTER)
    if rank is not None:
        kw.update(pgrid=pgrid[:dim], rank=rank)
    return workload.TGVSpec(**kw)


def _rank_setup(rank, G, dim, pgrid, n, singular, ordering=None):
    """particles, plan, context, volumes with forward comm, assembled Poisson system of one rank"""
    nreal = int(np.prod(pgrid[:dim]))
    if rank < nreal:
        spec = _spec(dim, pgrid, n, rank)
        parts = dist.prune_ghosts(workload.make_tgv(spec))     # column map = the referenced tags only, like Epetra's
    else:                                               # an extra rank without particles
        spec = _spec(dim, pgrid, n, 0)
        parts = empty_parts(spec, rank)
    plan = dist.make_plan(parts, G.td(rank))
    ctx = G.context(rank, ordering=ordering)
    nl = int(parts["nlocal"])
    fwd = hip.HaloForward(ctx, nl, plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr)
    vf = hip.compute_volumes(ctx, parts, plan.colmap)
    vfrac = np.ascontiguousarray(dist.forward_scalar_rccl(fwd, plan, vf))          # [nall], ghosts from their owners
    fwd.close()
    A, b = hip.assemble_poisson(ctx, parts, plan.colmap, spec.dt, parts["rho"], np.ascontiguousarray(parts["v"]), vfrac=vfrac,
                                ncol=plan.ncol, singular=singular, rank0=(rank == 0))
    if plan.ncol > nl:
        A.set_halo(plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr)
    rp, ci, v = A.export_csr()
    col_tag = np.zeros(plan.ncol, dtype=np.int64)
    col_tag[plan.colmap] = parts["tag"]
    return dict(spec=spec, parts=parts, plan=plan, ctx=ctx, A=A, b=b, csr=(rp, ci, v), col_tag=col_tag, nl=nl,
                rtag=parts["tag"][:nl].astype(np.int64))


class GlobalOracle:
    """the single-rank oracle system of the whole lattice and its permutation to rank-concatenated order"""

    def __init__(self, dim, pgrid, n, singular, rtags):
        spec = _spec(dim, pgrid, n)
        parts = workload.make_tgv(spec)
        P = orc.Particles(parts, workload.single_rank_colmap(parts)).precompute(corrections=False)
        rp, ci, val, b = P.poisson(spec.dt, parts["rho"], parts["v"], antisym=True, singular=singular)
        self.N = N = parts["nlocal"]
        self.A = sps.csr_matrix((val, ci, rp), shape=(N, N))
        self.b = b
        self.scale = np.abs(val).max()
        self.pos = np.zeros(N + 1, dtype=np.int64)
        self.pos[parts["tag"][:N].astype(np.int64)] = np.arange(N)
        self.perm = np.concatenate([self.pos[t] for t in rtags])            # concatenated row -> oracle row
        assert len(np.unique(self.perm)) == N
        self.off = np.concatenate([[0], np.cumsum([len(t) for t in rtags])]).astype(np.int64)
        Ap = self.A[self.perm][:, self.perm].tocsr()
        Ap.sort_indices()
        self.Ap = Ap
        self.bp = b[self.perm]

    def check_rows(self, results):
        """every rank's device-assembled rows and right-hand side against the oracle's, keyed by global tags"""
        for r, res in enumerate(results):
            rp, ci, v = res["csr"]
            nl = res["nl"]
            if nl == 0:
                continue
            rows = np.repeat(self.pos[res["rtag"]], np.diff(rp))
            D = sps.csr_matrix((v, (rows, self.pos[res["col_tag"][ci]])), shape=(self.N, self.N))
            sel = self.pos[res["rtag"]]
            assert abs(D[sel] - self.A[sel]).max() <= 1e-12 * self.scale, "rank %d: assembled rows differ" % r
            assert np.max(np.abs(res["b"] - self.b[sel])) <= 1e-12 * np.abs(self.b).max(), "rank %d: rhs differs" % r

    def block_ptr(self, block):
        bp = [0]
        for r in range(len(self.off) - 1):
            lo, hi = int(self.off[r]), int(self.off[r + 1])
            bp += list(range(lo + block, hi, block)) + ([hi] if hi > lo else [])
        return np.asarray(bp, dtype=np.int32)


def _solve_bjacobi(rank, G, dim, pgrid, n, singular):
    st = _rank_setup(rank, G, dim, pgrid, n, singular)
    ctx, A = st["ctx"], st["A"]
    try:
        M = hip.Precond(ctx, A, "bjacobi-ilu0", BLOCK)
        x, bb = np.zeros(st["nl"]), st["b"].copy()
        info = hip.solve(ctx, A, bb, x, prec=M, singular=(singular == orc.NULLSPACE))
        # one more product through the halo path, for the explicit residual of the global system
        y = A.spmv(x)
        M.close()
        return dict(st, x=x, y=y, bproj=bb, info=(info.converged, info.iters, info.rel_res_explicit), ctx=None, A=None, parts=None)
    finally:
        A.close()
        ctx.close()


@pytest.mark.parametrize("dim,pgrid,n", [(3, (2, 1, 1), 8), (3, (2, 2, 1), 8), (3, (2, 2, 2), 8), (2, (2, 2, 1), 16),
                                         (3, (1, 2, 2), 8)])
def test_distributed_fgmres_bjacobi_ilu0_matches_the_single_rank_oracle(dim, pgrid, n):
    """2x1x1, 2x2x1 and 2x2x2 bricks (the 2-, 4- and 8-GPU grids of bench.py; 2x2x2 is the decomposition of BASELINE
    configs[2]): FGMRES(50) + block-Jacobi ILU(0) with the NullSpace projection over all ranks."""
    world = int(np.prod(pgrid[:dim]))
    G = RankGroup(world)
    try:
        res = G.run(_solve_bjacobi, dim, pgrid, n, orc.NULLSPACE)
        cnt = G.counts()
    finally:
        G.close()
    O = GlobalOracle(dim, pgrid, n, orc.NULLSPACE, [r["rtag"] for r in res])
    O.check_rows(res)
    # every rank has peers that are not itself, and not all ranks have the same peers
    peers = [tuple(int(p) for p in r["plan"].peers) for r in res]
    assert all(len(p) > 0 and q not in p for q, p in enumerate(peers)), peers
    if world > 2:
        assert len(set(peers)) > 1
    ilu = orc.ILU(O.Ap.indptr, O.Ap.indices, O.Ap.data, 0, O.block_ptr(BLOCK))
    xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=True, prec="ilu", ilu=ilu)
    infos = {r["info"][:2] for r in res}
    assert len(infos) == 1, "ranks disagree on convergence / iteration count: %s" % infos
    conv, iters = infos.pop()
    assert conv == 1 and io.converged == 1 and abs(iters - io.iters) <= 1, (iters, io.iters)
    x = np.concatenate([r["x"] for r in res])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
    # explicit residual of the GLOBAL system from the distributed product, projected like the operator
    y = np.concatenate([r["y"] for r in res])
    bb = np.concatenate([r["bproj"] for r in res])
    rres = bb - (y - y.mean())
    assert np.linalg.norm(rres) / np.linalg.norm(bb) < 2e-8
    assert np.max(np.abs(y - O.Ap @ x)) <= 1e-11 * O.scale * np.abs(x).max()     # halo SpMV == global operator
    assert abs(x.mean()) <= 1e-12 * np.abs(x).max()
    assert cnt["exchanges"] > world * iters and cnt["allreduces"] > world * iters
    print("ranks %d grid %s: iterations %d (oracle %d), %d exchanges, %d all-reduces" % (world, pgrid, iters, io.iters,
                                                                                         cnt["exchanges"], cnt["allreduces"]))


def _solve_on_library_bricks(rank, G, dim, pgrid, n):
    st = _rank_setup(rank, G, dim, pgrid, n, orc.NULLSPACE, ordering="bricks")
    ctx, A = st["ctx"], st["A"]
    try:
        o = A.ordering()
        M = hip.Precond(ctx, A, "bjacobi-ilu0", 0)             # every rank's own bricks
        x, bb = np.zeros(st["nl"]), st["b"].copy()
        info = hip.solve(ctx, A, bb, x, prec=M, singular=True)
        y = A.spmv(x)
        M.close()
        return dict(st, x=x, y=y, bproj=bb, perm=o["perm"], bptr=o["block_ptr"], info=(info.converged, info.iters), ctx=None, A=None, parts=None)
    finally:
        A.close()
        ctx.close()


@pytest.mark.parametrize("pgrid,n", [((2, 1, 1), 10), ((2, 2, 1), 10)])
def test_distributed_solve_in_the_librarys_row_numbering(pgrid, n):
    """isph_ctx_set_ordering(BRICKS) on more than one rank: every rank sorts ITS particles into bricks, ghost columns keep the
    numbering of the halo plan and isph_mat_set_halo translates the send list.  Rows exported in the caller's numbering =
    the single-rank oracle's; the oracle is handed every rank's permutation and subdomain table (block-diagonal P over the
    rank-concatenated system) and must need the same iterations and give the same pressure."""
    dim = 3
    world = int(np.prod(pgrid))
    G = RankGroup(world)
    try:
        res = G.run(_solve_on_library_bricks, dim, pgrid, n)
    finally:
        G.close()
    O = GlobalOracle(dim, pgrid, n, orc.NULLSPACE, [r["rtag"] for r in res])
    O.check_rows(res)                                             # exported in the caller's numbering, ghost columns by tag
    gperm = np.concatenate([O.off[r] + res[r]["perm"].astype(np.int64) for r in range(world)])
    gbp = np.concatenate([[0]] + [O.off[r] + res[r]["bptr"][1:].astype(np.int64) for r in range(world)]).astype(np.int32)
    assert all(np.diff(res[r]["bptr"]).max() <= 1024 for r in range(world))
    App = O.Ap[gperm][:, gperm].tocsr()
    App.sort_indices()
    ilu = orc.ILU(App.indptr, App.indices, App.data, 0, gbp)
    xoi, io, _ = orc.solve(App.indptr, App.indices, App.data, O.bp[gperm], singular=True, prec="ilu", ilu=ilu)
    xo = np.empty(O.N)
    xo[gperm] = xoi
    infos = {r["info"] for r in res}
    assert len(infos) == 1, infos
    conv, iters = infos.pop()
    assert conv == 1 and io.converged == 1 and abs(iters - io.iters) <= 1, (iters, io.iters)
    x = np.concatenate([r["x"] for r in res])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
    y = np.concatenate([r["y"] for r in res])
    assert np.max(np.abs(y - O.Ap @ x)) <= 1e-11 * O.scale * np.abs(x).max()     # halo SpMV in the caller's numbering == global operator
    bb = np.concatenate([r["bproj"] for r in res])
    rres = bb - (y - y.mean())
    assert np.linalg.norm(rres) / np.linalg.norm(bb) < 2e-8


@pytest.mark.parametrize("mode", [orc.PINZERO, orc.DOUBLEDIAG])
def test_rank0_only_singular_modes_across_ranks(mode):
    """PinZero / DoubleDiag modify ONE row, on rank 0 only (modifySingularMatrix, pair_isph.cpp:493-520, called once on
    rank 0 by functor_incomp_navier_stokes_poisson.h:150-170): the other ranks must assemble plain rows, and the solve
    (no projection) must land on the oracle's solution of the same modified system."""
    dim, pgrid, n = 3, (2, 2, 1), 8
    G = RankGroup(4)
    try:
        res = G.run(_solve_bjacobi, dim, pgrid, n, mode)
    finally:
        G.close()
    O = GlobalOracle(dim, pgrid, n, mode, [r["rtag"] for r in res])
    O.check_rows(res)
    Oplain = GlobalOracle(dim, pgrid, n, orc.NULLSPACE, [r["rtag"] for r in res])
    changed = np.unique((abs(O.Ap - Oplain.Ap)).tocoo().row)
    assert len(changed) == 1 and changed[0] < O.off[1], "exactly one row, owned by rank 0, differs from the plain operator"
    ilu = orc.ILU(O.Ap.indptr, O.Ap.indices, O.Ap.data, 0, O.block_ptr(BLOCK))
    xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=False, prec="ilu", ilu=ilu)
    infos = {r["info"][:2] for r in res}
    assert len(infos) == 1
    conv, iters = infos.pop()
    assert conv == 1 and io.converged == 1 and abs(iters - io.iters) <= 1, (iters, io.iters)
    x = np.concatenate([r["x"] for r in res])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6


def test_a_rank_without_particles_takes_part_in_the_collectives():
    """LAMMPS allows empty subdomains: the rank owns no rows, has no peers, and still enters every all-reduce of the
    Krylov loop (a rank that skipped them would hang the others).  3 ranks: 2x1x1 bricks + one empty rank."""
    dim, pgrid, n = 3, (2, 1, 1), 8
    G = RankGroup(3)
    try:
        res = G.run(_solve_bjacobi, dim, pgrid, n, orc.NULLSPACE)
    finally:
        G.close()
    assert res[2]["nl"] == 0 and len(res[2]["plan"].peers) == 0
    O = GlobalOracle(dim, pgrid, n, orc.NULLSPACE, [r["rtag"] for r in res])
    O.check_rows(res)
    ilu = orc.ILU(O.Ap.indptr, O.Ap.indices, O.Ap.data, 0, O.block_ptr(BLOCK))
    xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=True, prec="ilu", ilu=ilu)
    infos = {r["info"][:2] for r in res}
    assert len(infos) == 1
    conv, iters = infos.pop()
    assert conv == 1 and abs(iters - io.iters) <= 1
    x = np.concatenate([r["x"] for r in res])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6


# ------------------------------------------------------------------ "Overlap Level" 1 across ranks
def _solve_overlap(rank, G, dim, pgrid, n, fill, combine, levels=1):
    st = _rank_setup(rank, G, dim, pgrid, n, orc.NULLSPACE)
    ctx, A, plan = st["ctx"], st["A"], st["plan"]
    try:
        rp, ci, v = st["csr"]
        if levels == 1:
            rpe, cie, ve = dist.extend_rows(plan, rp, ci, v, G.td(rank))
            xplan = plan
        else:                               # L layers: halo triples per layer and owner, non-neighbour ranks included
            rpe, cie, ve, xplan = dist.extend_rows_levels(plan, rp, ci, v, G.td(rank), levels=levels)
        Aext = hip.Matrix.from_csr(ctx, rpe, cie, ve)
        M = hip.PrecondOverlap(ctx, Aext, xplan, level_of_fill=fill, combine=combine)
        Aext.close()
        r = np.cos(0.37 * st["rtag"].astype(np.float64))
        z = M.apply(r)
        x, bb = np.zeros(st["nl"]), st["b"].copy()
        info = hip.solve(ctx, A, bb, x, prec=M, singular=True)
        M.close()
        return dict(st, x=x, r=r, z=z, info=(info.converged, info.iters), ctx=None, A=None, parts=None, xpeers=[int(p) for p in xplan.peers],
                    next=len(rpe) - 1)
    finally:
        A.close()
        ctx.close()


@pytest.mark.parametrize("levels,fill,combine", [(2, 0, "add"), (3, 0, "add"), (2, 1, "zero")])
def test_overlap_levels_above_one_across_ranks(levels, fill, combine):
    """"Overlap Level" L > 1 on several ranks (precond_ifpack.h:43): four slabs of 8 cells, so the third layer reaches a
    rank the matrix' own halo never talks to.  Layers gathered by dist.extend_rows_levels, factored and applied by
    isph_prec_create_overlap with one (peer, send, receive) triple per layer and owner; against
    oracle/isph_schwarz_oracle.c with overlap = L on the global matrix (application 1e-10, iterations +-1, x 1e-6)."""
    dim, pgrid, n = 3, (4, 1, 1), 8
    G = RankGroup(4)
    try:
        res = G.run(_solve_overlap, dim, pgrid, n, fill, combine, levels)
    finally:
        G.close()
    O = GlobalOracle(dim, pgrid, n, orc.NULLSPACE, [r["rtag"] for r in res])
    O.check_rows(res)
    S = orc.Schwarz(O.Ap.indptr, O.Ap.indices, O.Ap.data, fill, own_ptr=O.off.astype(np.int32), overlap=levels, combine=combine)
    rows, lp, _, _, _ = S.export()
    assert [q["next"] for q in res] == [int(lp[k + 1] - lp[k]) for k in range(4)]          # the same extended subdomains
    if levels >= 3:
        assert any(len(set(q["xpeers"])) == 3 for q in res), "a third layer must import rows of the non-neighbour slab"
    r = np.concatenate([q["r"] for q in res])
    z = np.concatenate([q["z"] for q in res])
    zo = S.apply(r)
    assert np.max(np.abs(z - zo)) <= 1e-10 * np.abs(zo).max()
    xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=True, prec="schwarz", schwarz=S)
    infos = {q["info"] for q in res}
    assert len(infos) == 1
    conv, iters = infos.pop()
    assert conv == 1 and io.converged == 1 and abs(iters - io.iters) <= 1, (iters, io.iters)
    x = np.concatenate([q["x"] for q in res])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
    print("overlap-%d ILU(%d) %s on 4 slabs: iterations %d (oracle %d), extended rows %s" % (levels, fill, combine, iters, io.iters,
                                                                                            [q["next"] for q in res]))


@pytest.mark.parametrize("pgrid,fill,combine", [((2, 1, 1), 0, "add"), ((2, 2, 1), 1, "add"), ((2, 2, 1), 0, "zero"),
                                                ((2, 2, 2), 0, "add")])
def test_overlap_one_schwarz_across_different_owners(pgrid, fill, combine):
    """Ifpack_AdditiveSchwarz<ILU(k)> with "Overlap Level" 1 on several ranks (the reference's default decomposition,
    precond_ifpack.h:35-43,60-74): gather of the ghost part of r from its OWNERS, ILU(k) of the extended subdomain,
    and with combine Add the return of the ghost part of z to a different rank, which adds it.  One application and
    the whole solve against oracle/isph_schwarz_oracle.c on the global matrix with one subdomain per rank."""
    dim, n = 3, 8
    world = int(np.prod(pgrid))
    G = RankGroup(world)
    try:
        res = G.run(_solve_overlap, dim, pgrid, n, fill, combine)
    finally:
        G.close()
    O = GlobalOracle(dim, pgrid, n, orc.NULLSPACE, [r["rtag"] for r in res])
    O.check_rows(res)
    S = orc.Schwarz(O.Ap.indptr, O.Ap.indices, O.Ap.data, fill, own_ptr=O.off.astype(np.int32), overlap=1, combine=combine)
    r = np.concatenate([q["r"] for q in res])
    z = np.concatenate([q["z"] for q in res])
    zo = S.apply(r)
    assert np.max(np.abs(z - zo)) <= 1e-10 * np.abs(zo).max()
    xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=True, prec="schwarz", schwarz=S)
    infos = {q["info"] for q in res}
    assert len(infos) == 1
    conv, iters = infos.pop()
    assert conv == 1 and io.converged == 1 and abs(iters - io.iters) <= 1, (iters, io.iters)
    x = np.concatenate([q["x"] for q in res])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
    print("overlap-1 ILU(%d) %s on %d ranks: iterations %d (oracle %d)" % (fill, combine, world, iters, io.iters))


# ------------------------------------------------------------------ the C++ mirror on several MPI ranks sharing the GPU
def _export_rank(rank, G, dim, pgrid, n):
    st = _rank_setup(rank, G, dim, pgrid, n, orc.NULLSPACE)
    st["A"].close()
    st["ctx"].close()
    return dict(st, ctx=None, A=None, parts=None)


@pytest.mark.parametrize("pgrid,mode", [((2, 1, 1), "bjacobi"), ((2, 2, 1), "bjacobi"), ((2, 2, 1), "overlap"), ((2, 1, 1), "ml"),
                                        ((4, 1, 1), "overlap2")])
def test_cpp_mirror_on_mpi_ranks_sharing_one_gpu(tmp_path, pgrid, mode):
    """SolverLin(MPI_Comm&) on 2 and 4 real MPI ranks (`mpiexec -n N`, fresh processes), all on device 0: the mirror
    sees that the ranks share a device and takes the MPI transport (host/mpi_transport.h: MPI_Isend/Irecv/Waitall for
    the Import, MPI_Allreduce for the dots) in place of RCCL; matrix ingress with the lists of each rank's
    Epetra_Import (host/halo_lists.h), PrecondWrapper_Ifpack on 256-row blocks / with "Overlap Level" 1 (row import
    over MPI_Sendrecv), PrecondWrapper_ML with the null vector.  Against the single-rank oracle of the whole system."""
    import subprocess
    from isph_amd import build
    exes = build.build_cpp_mpi()
    if exes is None:
        pytest.skip("no MPI installation (mpi.h / mpiexec / libmpi) on this machine")
    dim, n = 3, 8
    world = int(np.prod(pgrid))
    G = RankGroup(world)
    try:
        res = G.run(_export_rank, dim, pgrid, n)
    finally:
        G.close()
    O = GlobalOracle(dim, pgrid, n, orc.NULLSPACE, [r["rtag"] for r in res])
    O.check_rows(res)
    for rank, r in enumerate(res):
        plan = r["plan"]
        rp, ci, v = r["csr"]
        nsend, nrecv = np.diff(plan.send_ptr), np.diff(plan.recv_ptr)
        to = [k for k in range(plan.npeers) if nsend[k] > 0]
        frm = [k for k in range(plan.npeers) if nrecv[k] > 0]
        with open(tmp_path / ("rank%d.bin" % rank), "wb") as f:
            np.array([r["nl"], plan.ncol, len(v)], np.int32).tofile(f)
            rp.astype(np.int32).tofile(f); ci.astype(np.int32).tofile(f); v.tofile(f); r["b"].tofile(f)
            np.array([len(to)], np.int32).tofile(f)
            plan.peers[to].astype(np.int32).tofile(f); nsend[to].astype(np.int32).tofile(f)
            np.array([int(plan.send_ptr[-1])], np.int32).tofile(f)
            plan.send_idx.astype(np.int32).tofile(f)
            np.array([len(frm)], np.int32).tofile(f)
            plan.peers[frm].astype(np.int32).tofile(f); nrecv[frm].astype(np.int32).tofile(f)
    run = subprocess.run([build.MPIEXEC, "-n", str(world), exes[0], str(tmp_path), "-", "1", "ranks", mode],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ranks share a device" in run.stdout
    xs, its = [], set()
    for rank in range(world):
        with open(tmp_path / ("x%d.bin" % rank), "rb") as f:
            conv, iters = np.fromfile(f, np.int32, 2)
            xs.append(np.fromfile(f, np.float64))
        assert conv == 1
        its.add(int(iters))
    assert len(its) == 1
    iters = its.pop()
    x = np.concatenate(xs)
    if mode == "bjacobi":
        ilu = orc.ILU(O.Ap.indptr, O.Ap.indices, O.Ap.data, 0, O.block_ptr(BLOCK))
        xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=True, prec="ilu", ilu=ilu)
    elif mode.startswith("overlap"):
        S = orc.Schwarz(O.Ap.indptr, O.Ap.indices, O.Ap.data, 0, own_ptr=O.off.astype(np.int32), overlap=int(mode[7:] or 1), combine="add")
        xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=True, prec="schwarz", schwarz=S)
    else:
        xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=True, prec="jacobi")
    assert io.converged == 1
    if mode != "ml":
        assert abs(iters - io.iters) <= 1, (iters, io.iters)
    else:
        assert iters < io.iters, (iters, io.iters)      # the rank-local hierarchies must beat point Jacobi on the same system
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
    print("mpiexec -n %d %s: iterations %d (oracle %s %d)" % (world, mode, iters, "jacobi" if mode == "ml" else mode, io.iters))


# ------------------------------------------------------------------ rank-local SA-AMG across ranks
AMG_KW = dict(theta=0.02, block=256, coarse_max=64)


def _solve_amg(rank, G, dim, pgrid, n, mode, sweeps, kw=None, ordering=None):
    st = _rank_setup(rank, G, dim, pgrid, n, mode, ordering=ordering)
    ctx, A = st["ctx"], st["A"]
    try:
        nullvec = None
        if mode == orc.NULLSPACE:                                             # setNullVector: the normalised mask of ALL ranks
            nullvec = np.full(st["nl"], 1.0 / np.sqrt(float(np.prod(pgrid[:dim])) * n ** dim))
        M = hip.PrecondAMG(ctx, A, nullvec=nullvec, params=hip.AmgParams(sweeps=sweeps, **(kw or AMG_KW)))   # collective set-up
        levels = M.levels
        P0 = M.export(0, "P") if levels > 1 else None
        A1 = M.export(1, "A") if levels > 1 else None
        agg = M.aggregates(0) if levels > 1 else None
        r = np.cos(0.37 * st["rtag"].astype(np.float64))
        z = M.apply(r)                                                      # fine AND coarse residuals go through a halo
        x, bb = np.zeros(st["nl"]), st["b"].copy()
        info = hip.solve(ctx, A, bb, x, prec=M, singular=(mode == orc.NULLSPACE))
        M.close()
        return dict(st, levels=levels, P0=P0, A1=A1, agg=agg, r=r, z=z, x=x, info=(info.converged, info.iters), ctx=None, A=None,
                    parts=None)
    finally:
        A.close()
        ctx.close()


def _block_sgs(Ass, block, r):
    """z = M^-1 r with M = (D+L) D^-1 (D+U) of every diagonal block of `block` rows (the rank's Gauss-Seidel, local to
    the row blocks like ML's is local to the processor)"""
    import scipy.sparse.linalg as spla
    n = Ass.shape[0]
    z = np.zeros(n)
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        B = Ass[lo:hi][:, lo:hi].tocsr()
        d = B.diagonal()
        y = spla.spsolve_triangular(sps.tril(B, format="csr"), r[lo:hi], lower=True)
        z[lo:hi] = spla.spsolve_triangular(sps.triu(B, format="csr"), d * y, lower=False)
    return z


@pytest.mark.parametrize("mode,sweeps", [(orc.DOUBLEDIAG, 1), (orc.NULLSPACE, 2)])
def test_amg_coarse_levels_across_ranks(mode, sweeps):
    """PrecondWrapper_ML's stand-in on 4 ranks.  Aggregates and the smoothed prolongator stay on the rank (ML's Uncoupled
    aggregation, precond_ml.h:49); the Galerkin operator is P^T A P with the WHOLE A: coarse rows couple to the neighbours'
    aggregates and the coarse level exchanges a halo of its own (csrc/amg.hpp, amg_extend_prolongator).  Checked:
    (i) every rank's aggregates / P equal oracle/isph_amg_oracle.c on the rank's filtered matrix, entry by entry;
    (ii) every rank's coarse rows equal the rows of Pg^T A Pg (Pg = blockdiag of the ranks' P, A the GLOBAL oracle matrix):
    the owned block entry by entry, the ghost columns as the sorted values of the row's other entries;
    (iii) ONE application on all ranks equals the two-level cycle written out in numpy with that global coarse operator --
    DoubleDiag: the coarse systems of all ranks solved as one (the dense inverse every rank holds); NullSpace with the
    null vector, 2 sweeps: the coarse level smoothed twice, its residual through the coarse halo;
    (iv) FGMRES with it lands on the global solution in fewer iterations than with block-Jacobi ILU(0)."""
    dim, pgrid, n = 3, (2, 2, 1), 8
    G = RankGroup(4)
    try:
        res = G.run(_solve_amg, dim, pgrid, n, mode, sweeps)
    finally:
        G.close()
    O = GlobalOracle(dim, pgrid, n, mode, [q["rtag"] for q in res])
    O.check_rows(res)
    Pg, Ass, ncs = [], [], []
    for rank, q in enumerate(res):
        assert q["levels"] == 2, "the numpy cycle below is the two-level one"
        rp, ci, v = q["csr"]
        nl = q["nl"]
        keep = ci < nl                                                       # Ifpack_LocalFilter-like: owned columns only
        rpf = np.zeros(nl + 1, np.int32)
        rpf[1:] = np.cumsum(np.add.reduceat(keep.astype(np.int64), rp[:-1]))
        cif, vf = ci[keep], v[keep]
        nv = None if mode != orc.NULLSPACE else np.full(nl, 1.0 / np.sqrt(float(O.N)))
        Go = orc.AMG(rpf, cif, vf, nullvec=nv, **AMG_KW)
        assert np.array_equal(Go.aggregates(0), q["agg"])
        ro, co, vo = Go.export(0, "P")
        rg, cg, vg = q["P0"]
        assert np.array_equal(ro, rg) and np.array_equal(co, cg) and np.max(np.abs(vo - vg)) <= 1e-12 * np.abs(vo).max()
        nc = len(q["A1"][0]) - 1
        ncs.append(nc)
        Pg.append(sps.csr_matrix((vg, cg, rg), shape=(nl, nc)))
        Ass.append(sps.csr_matrix((vf, cif, rpf), shape=(nl, nl)))
    Pglob = sps.block_diag(Pg, format="csr")
    Ac = (Pglob.T @ O.Ap @ Pglob).tocsr()
    Ac.sort_indices()
    coff = np.concatenate([[0], np.cumsum(ncs)]).astype(np.int64)
    scale = np.abs(Ac.data).max()
    nghost = 0
    for rank, q in enumerate(res):
        rg, cg, vg = q["A1"]
        nc = ncs[rank]
        lo = int(coff[rank])
        D = sps.csr_matrix((vg, cg, rg), shape=(nc, int(cg.max()) + 1 if len(cg) else nc)).tocsr()
        own = D[:, :nc]
        assert abs(own - Ac[lo:lo + nc][:, lo:lo + nc]).max() <= 1e-11 * scale, "rank %d: owned block of the coarse operator" % rank
        ref_rows = Ac[lo:lo + nc].tolil()
        ref_rows[:, lo:lo + nc] = 0
        ref_rows = ref_rows.tocsr()
        ref_rows.eliminate_zeros()
        gh = D[:, nc:].tocsr()
        nghost += gh.nnz
        for i in range(nc):
            a = np.sort(gh.data[gh.indptr[i]:gh.indptr[i + 1]])
            b = np.sort(ref_rows.data[ref_rows.indptr[i]:ref_rows.indptr[i + 1]])
            a, b = a[np.abs(a) > 1e-13 * scale], b[np.abs(b) > 1e-13 * scale]
            assert len(a) == len(b) and np.max(np.abs(a - b), initial=0.0) <= 1e-11 * scale, "rank %d coarse row %d: couplings to the neighbours" % (rank, i)
    assert nghost > 0, "the coarse operator has no couplings across ranks"
    r = np.concatenate([q["r"] for q in res])
    sl = [slice(int(O.off[k]), int(O.off[k + 1])) for k in range(4)]
    csl = [slice(int(coff[k]), int(coff[k + 1])) for k in range(4)]
    blk = AMG_KW["block"]
    sgs = lambda rr: np.concatenate([_block_sgs(Ass[k], blk, rr[sl[k]]) for k in range(4)])
    x = sgs(r)                                                                                # pre-smoothing, zero guess
    for _ in range(1, sweeps):
        x = x + sgs(r - O.Ap @ x)
    bc = Pglob.T @ (r - O.Ap @ x)
    if mode == orc.NULLSPACE:                                                # the smoother as the coarse solver, 64-row blocks per rank
        Acc = [Ac[csl[k]][:, csl[k]].tocsr() for k in range(4)]
        sgc = lambda rr: np.concatenate([_block_sgs(Acc[k], 64, rr[csl[k]]) for k in range(4)])
        xc = sgc(bc)
        for _ in range(1, sweeps):
            xc = xc + sgc(bc - Ac @ xc)
    else:
        xc = np.linalg.solve(Ac.toarray(), bc)
    x = x + Pglob @ xc
    for _ in range(sweeps):
        x = x + sgs(r - O.Ap @ x)                                                            # post-smoothing
    z = np.concatenate([q["z"] for q in res])
    assert np.max(np.abs(z - x)) <= 1e-9 * np.abs(x).max()
    infos = {q["info"] for q in res}
    assert len(infos) == 1
    conv, iters = infos.pop()
    ilu = orc.ILU(O.Ap.indptr, O.Ap.indices, O.Ap.data, 0, O.block_ptr(BLOCK))
    xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=(mode == orc.NULLSPACE), prec="ilu", ilu=ilu)
    assert conv == 1 and iters < io.iters, (iters, io.iters)
    xs = np.concatenate([q["x"] for q in res])
    assert np.linalg.norm(xs - xo) / np.linalg.norm(xo) <= 1e-6
    print("AMG with coarse levels across 4 ranks: iterations %d (block-Jacobi ILU(0): %d)" % (iters, io.iters))


@pytest.mark.parametrize("mode,ordering", [(orc.DOUBLEDIAG, "caller"), (orc.NULLSPACE, "caller"), (orc.NULLSPACE, "bricks")])
def test_amg_deeper_hierarchy_across_eight_ranks(mode, ordering):
    """2x2x2 ranks, every rank with seven different peers, coarse_max small enough for three or more levels: every level
    below the fine one has its own halo plan, derived from the level above.  All ranks report the same depth, iteration
    count and convergence; the solution is the single-rank oracle's (ILU-preconditioned) to 1e-6.  ordering "bricks": the
    hierarchy is built on matrices in the library's own row numbering (every rank sorts its particles; the null vector and
    the vectors of the solve cross the C ABI in the caller's numbering)."""
    dim, pgrid, n = 3, (2, 2, 2), 16
    kw = dict(theta=0.02, block=256, coarse_max=8)
    G = RankGroup(8)
    try:
        res = G.run(_solve_amg, dim, pgrid, n, mode, 1, kw, ordering)
    finally:
        G.close()
    O = GlobalOracle(dim, pgrid, n, mode, [q["rtag"] for q in res])
    assert len({q["levels"] for q in res}) == 1 and res[0]["levels"] >= 3, [q["levels"] for q in res]
    infos = {q["info"] for q in res}
    assert len(infos) == 1, infos
    conv, iters = infos.pop()
    ilu = orc.ILU(O.Ap.indptr, O.Ap.indices, O.Ap.data, 0, O.block_ptr(BLOCK))
    xo, io, _ = orc.solve(O.Ap.indptr, O.Ap.indices, O.Ap.data, O.bp, singular=(mode == orc.NULLSPACE), prec="ilu", ilu=ilu)
    assert conv == 1 and io.converged == 1 and iters < io.iters, (iters, io.iters)
    xs = np.concatenate([q["x"] for q in res])
    rr = O.bp - O.Ap @ xs                                                   # explicit residual of the GLOBAL system
    if mode == orc.NULLSPACE:
        rr = rr - rr.mean()
    assert np.linalg.norm(rr) / np.linalg.norm(O.bp) <= 2e-8
    # (DoubleDiag leaves the system a doubled diagonal entry away from singular: x moves by 1e-5 between two solves at 1e-8)
    assert np.linalg.norm(xs - xo) / np.linalg.norm(xo) <= (1e-6 if mode == orc.NULLSPACE else 1e-4)
    print("AMG, %d levels across 8 ranks: iterations %d (block-Jacobi ILU(0): %d)" % (res[0]["levels"], iters, io.iters))


def test_amg_with_a_rank_that_cannot_coarsen():
    """Five ranks on a 2x2x1 decomposition: the fifth owns no particles.  It cannot coarsen, the others can: it passes its
    (empty) level on unchanged and stays in every collective step; the four others build the same hierarchy as without it
    (same depth, same iteration count as the 4-rank run) and land on the same solution."""
    dim, pgrid, n = 3, (2, 2, 1), 8
    runs = []
    for world in (4, 5):
        G = RankGroup(world)
        try:
            runs.append(G.run(_solve_amg, dim, pgrid, n, orc.NULLSPACE, 1))
        finally:
            G.close()
    four, five = runs
    assert five[4]["nl"] == 0
    assert {q["levels"] for q in five} == {four[0]["levels"]} and four[0]["levels"] >= 2
    assert {q["info"] for q in five} == {q["info"] for q in four} and four[0]["info"][0] == 1
    for a, b in zip(four, five[:4]):
        assert np.max(np.abs(a["x"] - b["x"])) <= 1e-12 * np.abs(a["x"]).max()


# ------------------------------------------------------------------ BASELINE configs[2] at its own size, on one GPU
def _config2_rank(rank, G, n, pgrid=(2, 2, 2), prec="bjacobi-ilu0"):
    import time
    # atoms in lexicographic order over the rank's brick (create_atoms); the library numbers the rows (the product default)
    spec = workload.TGVSpec(dim=3, ncell=tuple(n * g for g in pgrid), pgrid=pgrid, rank=rank, brick=(n, n, n), mode=workload.ADVECT)
    parts = dist.prune_ghosts(workload.make_tgv(spec))
    plan = dist.make_plan(parts, G.td(rank))
    ctx = G.context(rank, ordering="bricks")
    nl = int(parts["nlocal"])
    try:
        fwd = hip.HaloForward(ctx, nl, plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr)
        vf = hip.compute_volumes(ctx, parts, plan.colmap)
        vfrac = np.ascontiguousarray(dist.forward_scalar_rccl(fwd, plan, vf))
        fwd.close()
        A, b = hip.assemble_poisson(ctx, parts, plan.colmap, spec.dt, parts["rho"], np.ascontiguousarray(parts["v"]), vfrac=vfrac,
                                    ncol=plan.ncol, rank0=(rank == 0))
        A.set_halo(plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr)
        sub = np.diff(A.subdomains())
        assert len(sub) == nl // 500 and sub.min() == sub.max() == 500     # 10 x 10 x 5 bricks tile every rank's 100^3
        x, bb = np.zeros(nl), b.copy()
        t0 = time.perf_counter()
        if prec == "sa-amg":                                # setNullVector: the normalised mask of all ranks
            M = hip.PrecondAMG(ctx, A, nullvec=np.full(nl, 1.0 / np.sqrt(float(nl) * float(np.prod(pgrid)))))
        else:
            M = hip.Precond(ctx, A, "bjacobi-ilu0", 0)
        info = hip.solve(ctx, A, bb, x, prec=M, singular=True)
        wall = time.perf_counter() - t0
        y = A.spmv(x)                                  # distributed product for the explicit residual of the global system
        ones = A.spmv(np.ones(nl))                     # A 1 = 0 across all rank boundaries
        M.close(); A.close()
        return dict(nl=nl, nghost=int(plan.ncol - nl), npeers=int(plan.npeers), info=(info.converged, info.iters, info.restarts),
                    sum_x=float(x.sum()), rr=float(((bb - y) ** 2).sum()), sum_r=float((bb - y).sum()), bb2=float((bb ** 2).sum()),
                    ones_max=float(np.abs(ones).max()), amax=float(np.abs(y).max()), xmax=float(np.abs(x).max()), wall=wall)
    finally:
        ctx.close()


def test_config2_eight_bricks_of_one_million_particles_on_one_gpu():
    """BASELINE configs[2] at its own size: 3-D TGV, 200^3 = 8 M particles, 2x2x2 bricks of 100^3 -- eight rank threads on
    ONE MI355X through the host-staged transport (about 40 GB of device memory), every rank with seven different peers and
    ~170 k ghost columns.  No CPU oracle at this size: the checks are the size-independent ones -- every rank reports the
    same iteration count and convergence, the explicit residual of the GLOBAL system formed from the distributed product
    (projected like the operator) is below 2e-8, A 1 = 0 across every rank boundary, the solution has zero mean."""
    G = RankGroup(8, timeout_s=600.0)
    try:
        res = G.run(_config2_rank, 100)
        cnt = G.counts()
    finally:
        G.close()
    assert all(r["nl"] == 10 ** 6 and r["npeers"] == 7 and r["nghost"] > 100000 for r in res), [(r["nl"], r["npeers"], r["nghost"]) for r in res]
    infos = {r["info"] for r in res}
    assert len(infos) == 1, infos
    conv, iters, restarts = infos.pop()
    assert conv == 1 and 40 <= iters <= 200
    N = 8e6
    rr = sum(r["rr"] for r in res) - sum(r["sum_r"] for r in res) ** 2 / N        # || r - mean(r) ||^2 of the global residual
    bb2 = sum(r["bb2"] for r in res)
    assert np.sqrt(max(rr, 0.0) / bb2) < 2e-8
    assert max(r["ones_max"] for r in res) < 1e-9 * max(r["amax"] for r in res) / max(r["xmax"] for r in res) + 1e-6
    assert abs(sum(r["sum_x"] for r in res)) / N <= 1e-10 * max(r["xmax"] for r in res)
    print("configs[2] on one GPU: 8 x 10^6 rows, %d iterations (%d restarts), %.1f s per rank thread for set-up + solve, %d exchanges, %d all-reduces"
          % (iters, restarts, max(r["wall"] for r in res), cnt["exchanges"], cnt["allreduces"]))
