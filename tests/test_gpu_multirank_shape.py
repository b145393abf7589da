"""-m gpu: BASELINE configs[2]'s per-GPU shape and the multi-rank preconditioners through the ONE-GPU self-peer plan
(every periodic image is a ghost column received from rank 0 itself over RCCL): the N > 1 data path -- pack kernel,
grouped ncclSend/ncclRecv on the halo stream overlapped with the interior slices, ghost-column SpMV, all-reduced dots
-- on the 100^3 brick each of the 8 GPUs of configs[2] owns, and SA-AMG on a matrix that carries ghost columns."""
import numpy as np
import pytest
import scipy.sparse as sps

from isph_amd import dist, hip, workload
import oracle as orc
from problems import Problem, tgv_spec

pytestmark = pytest.mark.gpu


def _csr(rp, ci, v, n, m):
    return sps.csr_matrix((v, ci, rp), shape=(n, m))


def test_config2_brick_through_the_rccl_halo_path():
    """One 100^3 brick of BASELINE configs[2] (3-D TGV, 8 M particles on 2x2x2 GPUs = 10^6 rows per GPU) with its
    periodic images routed through the halo plan: 169 k ghost columns, interior / boundary slice split, exchange on
    the second stream.  Against the folded single-rank operator of the same brick: same right-hand side, iteration
    count within 1 (the dot products are reduced in another order), pressure vector <= 1e-6, and the residual
    re-computed on the HOST with an independent CSR product of the folded operator."""
    sp = tgv_spec(dim=3, n=100, mode=workload.ADVECT)
    p = workload.make_tgv(sp)
    n = p["nlocal"]
    ctx = hip.Context(0, rank=0, nranks=1, uid=hip.Context.unique_id())
    try:
        # folded operator (what bench.py --gpus 1 solves)
        colmap = workload.single_rank_colmap(p)
        vf = hip.compute_volumes(ctx, p, colmap)
        vfrac = np.ascontiguousarray(vf[p["owner_index"]])
        A0, b0 = hip.assemble_poisson(ctx, p, colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]), vfrac=vfrac)
        M0 = hip.Precond(ctx, A0, "bjacobi-ilu0", 512)
        x0 = np.zeros(n)
        i0 = hip.solve(ctx, A0, b0.copy(), x0, prec=M0, singular=True)
        rp, ci, v = A0.export_csr()
        M0.close(); A0.close()
        # the same brick with the images as ghost columns behind the RCCL exchange
        plan = dist.make_self_halo_plan(p)
        nghost = plan.ncol - n
        assert nghost > 150000 and plan.recv_ptr[-1] == nghost
        A, b = hip.assemble_poisson(ctx, p, plan.colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]), vfrac=vfrac, ncol=plan.ncol)
        A.set_halo(plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr)
        assert np.allclose(b, b0, rtol=0, atol=1e-12 * np.abs(b0).max())   # another summation order over the neighbours
        xr = np.random.default_rng(2).standard_normal(n)
        Ah = _csr(rp, ci, v, n, n)
        y = A.spmv(xr)
        yh = Ah @ xr
        assert np.max(np.abs(y - yh)) <= 1e-12 * np.abs(yh).max()        # ghost-column SpMV == folded operator
        M = hip.Precond(ctx, A, "bjacobi-ilu0", 512)
        x, bb = np.zeros(n), b.copy()
        info = hip.solve(ctx, A, bb, x, prec=M, singular=True)
        assert info.converged == 1 and i0.converged == 1 and abs(info.iters - i0.iters) <= 1, (info.iters, i0.iters)
        assert np.linalg.norm(x - x0) / np.linalg.norm(x0) <= 1e-6
        r = bb - Ah @ x
        r -= r.mean()
        assert np.linalg.norm(r) / np.linalg.norm(bb) < 2e-8
        assert abs(x.mean()) < 1e-12 * np.abs(x).max()
        M.close(); A.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("n,theta", [(16, 0.0), (20, 0.02)])
def test_amg_on_a_matrix_with_ghost_columns(n, theta):
    """isph_prec_create_amg on a matrix that carries ghost columns (self-peer plan): the hierarchy above the fine
    level is rank-local -- aggregation, prolongator and Galerkin products see the owned columns only (ML's Uncoupled
    aggregation; Ifpack_LocalFilter semantics), the fine-level smoother and residual see the halo.  So: aggregates, P and
    the coarse operators must be the oracle's for the matrix with the ghost columns dropped, entry by entry; and FGMRES
    with this preconditioner on the full operator converges to the folded system's solution, in an iteration count
    between the oracle's for the folded hierarchy and for the filtered matrix on every level."""
    pr = Problem(tgv_spec(dim=3, n=n, mode=workload.JITTER, brick=4))
    rp, ci, val, b = pr.poisson()                      # folded operator
    N = pr.n
    plan = dist.make_self_halo_plan(pr.parts)
    Ph = orc.Particles(pr.parts, plan.colmap)
    Ph.precompute(corrections=False)
    rph, cih, valh, bh = Ph.poisson(pr.spec.dt, pr.parts["rho"], pr.parts["v"], singular=orc.NULLSPACE)
    assert cih.max() >= N
    # local filter: ghost columns dropped
    keep = cih < N
    rpf = np.zeros(N + 1, np.int32)
    rpf[1:] = np.cumsum(np.add.reduceat(keep.astype(np.int64), rph[:-1]))
    cif, valf = cih[keep], valh[keep]
    nv = np.ones(N) / np.sqrt(N)
    kw = dict(theta=theta, block=256, coarse_max=64)
    G = orc.AMG(rpf, cif, valf, nullvec=nv, **kw)
    ctx = hip.Context(0, rank=0, nranks=1, uid=hip.Context.unique_id())
    try:
        A = hip.Matrix.from_csr(ctx, rph, cih, valh, ncol=plan.ncol)
        A.set_halo(plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr)
        M = hip.PrecondAMG(ctx, A, nullvec=nv, params=hip.AmgParams(**kw))
        assert M.levels == G.levels and M.levels >= 2
        for l in range(G.levels - 1):
            assert np.array_equal(G.aggregates(l), M.aggregates(l))
            ro, co, vo = G.export(l, "P")
            rg, cg, vg = M.export(l, "P")
            assert np.array_equal(ro, rg) and np.array_equal(co, cg)
            assert np.max(np.abs(vo - vg)) <= 1e-12 * np.abs(vo).max()
        for l in range(1, G.levels):
            ro, co, vo = G.export(l, "A")
            rg, cg, vg = M.export(l, "A")
            assert np.array_equal(ro, rg) and np.array_equal(co, cg)
            assert np.max(np.abs(vo - vg)) <= 1e-11 * np.abs(vo).max()
        x, bb = np.zeros(N), bh.copy()
        info = hip.solve(ctx, A, bb, x, prec=M, singular=True)
        # the oracle has two neighbours of this preconditioner: the hierarchy of the filtered matrix with the FILTERED
        # matrix on the fine level too (weaker: its smoother and residual miss the ghost couplings), and the hierarchy of
        # the folded operator (stronger: aggregates cross the periodic seam).  The device's iteration count must lie
        # between the two; the solution is the folded system's.
        xo, io, _ = orc.solve(rp, ci, val, b, singular=True, prec="amg", amg=G)
        Gfull = orc.AMG(rp, ci, val, nullvec=nv, **kw)
        _, iof, _ = orc.solve(rp, ci, val, b, singular=True, prec="amg", amg=Gfull)
        assert info.converged == 1 and io.converged == 1 and iof.converged == 1
        assert iof.iters - 1 <= info.iters <= io.iters + 1, (iof.iters, info.iters, io.iters)
        print("AMG with ghost columns: iterations folded-hierarchy %d <= device %d <= filtered everywhere %d" % (iof.iters, info.iters, io.iters))
        assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
        M.close(); A.close()
    finally:
        ctx.close()
