"""-m gpu: the row numbering the library owns (csrc/order.hpp, isph_ctx_set_ordering(ISPH_ORDER_BRICKS), the default).

The reference's rows follow LAMMPS' atom order (pair_isph.cpp:1258-1259); the library sorts a rank's owned particles into
bricks itself and shows none of it at the C ABI.  Checked here, for the SAME particles handed over in three atom orders
(the generator's 8^3 bricks, lexicographic = what create_atoms gives, shuffled = after migration):
  * the exported matrix and b are the oracle's for THAT atom order: pattern bit for bit, values <= 1e-12 max|A|;
  * the permutation is the stable sort of the documented key (oracle/order.py restates it from the reported geometry), the
    subdomain table covers the rows with blocks of 1..1024, and the internal matrix is the same for all three orders;
  * block ILU(0) on the library's bricks (isph_prec_create block_size 0): factor = orc.ILU on P A P^T with the same
    table (pattern exact, values 1e-10), FGMRES iterations +-1 and x <= 1e-6 against the oracle solving the permuted
    system, and the same iteration count whatever the atom order;
  * isph_spmv / isph_prec_apply / isph_solve take and return vectors in the caller's numbering;
  * PinZero / DoubleDiag modify the caller's first fluid row (modifySingularMatrix, pair_isph.cpp:493-520)."""
import numpy as np
import pytest
import scipy.sparse as sps

from isph_amd import hip, workload
import oracle as orc
import order as oorder
from problems import Problem, tgv_spec, wall_types

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def octx():
    ctx = hip.Context(0, ordering="bricks")
    yield ctx
    ctx.close()


def _three_orders(spec):
    """the same particle set in three atom orders: (name, parts)"""
    base = workload.make_tgv(spec)
    n = base["nlocal"]
    tag0 = base["tag"][:n].astype(np.int64) - 1
    lex = np.argsort(tag0, kind="stable")                       # tag = lattice site, x fastest: create_atoms order
    shuf = np.random.default_rng(7).permutation(n)
    return [("generator", base), ("lexicographic", workload.renumber(base, lex)), ("shuffled", workload.renumber(base, shuf))]


def _oracle_system(parts, spec, singular=orc.NULLSPACE, kinds=None):
    P = orc.Particles(parts, workload.single_rank_colmap(parts), kernel=spec.kernel, kinds=kinds).precompute(corrections=False)
    return P, P.poisson(spec.dt, parts["rho"], parts["v"], antisym=True, singular=singular)


def _assemble(ctx, parts, spec, P, singular=hip.NULLSPACE, kinds=None):
    colmap = workload.single_rank_colmap(parts)
    return hip.assemble_poisson(ctx, parts, colmap, spec.dt, parts["rho"], np.ascontiguousarray(parts["v"]), vfrac=P.vfrac,
                                singular=singular, kinds=kinds)


@pytest.mark.parametrize("dim,n", [(3, 20), (2, 48)])
def test_matrix_in_the_callers_numbering_whatever_the_atom_order(octx, dim, n):
    spec = tgv_spec(dim=dim, n=n, mode=workload.JITTER)
    internal = []
    for name, parts in _three_orders(spec):
        P, (rp, ci, val, b) = _oracle_system(parts, spec)
        A, bg = _assemble(octx, parts, spec, P)
        rp2, ci2, v2 = A.export_csr()
        assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci), name
        assert np.max(np.abs(v2 - val)) <= 1e-12 * np.abs(val).max(), name
        assert np.max(np.abs(bg - b)) <= 1e-12 * np.abs(b).max(), name
        o = A.ordering()
        assert o is not None
        nl = parts["nlocal"]
        # the permutation is the stable sort of the documented key; the table covers the rows
        perm = oorder.order(parts["x"][:nl], o["geom"], o["faces"])
        assert np.array_equal(perm, o["perm"]), name
        bp = oorder.block_table(parts["x"][:nl], o["geom"], o["faces"], perm)
        assert np.array_equal(bp, o["block_ptr"]), name
        sizes = np.diff(bp)
        assert bp[0] == 0 and bp[-1] == nl and sizes.min() >= 1 and sizes.max() <= 1024
        # the geometry follows the rule (spacing of the lattice, bricks of 10 x 10 x 5 / 22 x 22 cells at most 20 % over)
        g = oorder.geometry(parts["x"][:nl], dim)
        assert list(g.ncell[:dim]) == [n] * dim == list(o["geom"].ncell)[:dim]
        assert list(g.cells_per_brick) == list(o["geom"].cells_per_brick) and list(g.nbrick) == list(o["geom"].nbrick)
        assert np.allclose(g.lo[:dim], list(o["geom"].lo)[:dim], rtol=0, atol=1e-12)
        # the cell faces are the quantile rule applied to the histogram of the reported grid
        fr = oorder.faces_from_histogram(parts["x"][:nl], o["geom"])
        for a in range(dim):
            assert np.array_equal(fr[a], o["faces"][a]), (name, a)
            assert np.all(np.diff(o["faces"][a]) >= 0)
        # rows by position: the internal matrix does not depend on the atom order it was handed in
        rpi, cii, vi, _ = oorder.permute_system(rp, ci, val, None, o["perm"])
        internal.append((name, rpi, cii, vi, parts["x"][:nl][o["perm"]]))
        # x / y of isph_spmv are the caller's
        xv = np.random.default_rng(3).standard_normal(nl)
        yo = sps.csr_matrix((val, ci, rp), shape=(nl, nl)) @ xv
        assert np.max(np.abs(A.spmv(xv) - yo)) <= 1e-12 * np.abs(yo).max(), name
        A.close()
    for name, rpi, cii, vi, xi in internal[1:]:
        assert np.array_equal(xi, internal[0][4]), name                   # same particles on the same internal rows
        assert np.array_equal(rpi, internal[0][1]) and np.array_equal(cii, internal[0][2]), name
        assert np.max(np.abs(vi - internal[0][3])) <= 1e-12 * np.abs(vi).max(), name


def test_block_ilu_on_the_librarys_bricks_matches_the_oracle(octx):
    spec = tgv_spec(dim=3, n=20, mode=workload.JITTER)
    its = {}
    for name, parts in _three_orders(spec):
        P, (rp, ci, val, b) = _oracle_system(parts, spec)
        nl = parts["nlocal"]
        A, bg = _assemble(octx, parts, spec, P)
        o = A.ordering()
        M = hip.Precond(octx, A, "bjacobi-ilu0", 0)                       # block_size 0: the matrix' own subdomains
        assert M.info()["nblocks"] == len(o["block_ptr"]) - 1
        # the oracle gets the same permutation and the same table explicitly
        rpi, cii, vi, bi = oorder.permute_system(rp, ci, val, b, o["perm"])
        ref = orc.ILU(rpi, cii, vi, 0, o["block_ptr"])
        frp, fci, fv = ref.export()
        grp, gci, gv = M.export_ilu()                                      # internals: the matrix' numbering
        assert np.array_equal(grp, frp) and np.array_equal(gci, fci), name
        assert np.max(np.abs(gv - fv) / np.maximum(np.abs(fv), 1e-10 * np.abs(fv).max())) < 1e-10, name
        # r / z of isph_prec_apply are the caller's
        r = np.random.default_rng(5).standard_normal(nl)
        zo = np.empty(nl)
        zo[o["perm"]] = ref.apply(r[o["perm"]])
        z = M.apply(r)
        assert np.linalg.norm(z - zo) / np.linalg.norm(zo) < 1e-11, name
        # b, x of isph_solve are the caller's; the oracle solves the permuted system
        xoi, io, _ = orc.solve(rpi, cii, vi, bi, singular=True, prec="ilu", ilu=ref)
        xo = np.empty(nl)
        xo[o["perm"]] = xoi
        bb, xg = bg.copy(), np.zeros(nl)
        info = hip.solve(octx, A, bb, xg, prec=M, singular=True)
        assert info.converged == 1 and abs(info.iters - io.iters) <= 1, (name, info.iters, io.iters)
        assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) <= 1e-6, name
        assert abs(xg.sum()) <= 1e-10 * np.abs(xg).sum()                   # x . n = 0
        # b comes back projected, in the caller's numbering
        assert np.max(np.abs(bb - (b - b.mean()))) <= 1e-12 * np.abs(b).max(), name
        its[name] = info.iters
        M.close(); A.close()
    assert len(set(its.values())) == 1, its                                # the atom order does not reach the solver


def test_iluk_on_the_librarys_bricks(octx):
    """"bjacobi-ilu1" with block_size 0: the reference's default level of fill (precond_ifpack.h:35) on the library's bricks"""
    spec = tgv_spec(dim=3, n=20, mode=workload.JITTER)
    name, parts = _three_orders(spec)[1]
    P, (rp, ci, val, b) = _oracle_system(parts, spec)
    nl = parts["nlocal"]
    A, bg = _assemble(octx, parts, spec, P)
    o = A.ordering()
    M = hip.Precond(octx, A, "bjacobi-ilu1", 0)
    rpi, cii, vi, bi = oorder.permute_system(rp, ci, val, b, o["perm"])
    ref = orc.ILU(rpi, cii, vi, 1, o["block_ptr"])
    frp, fci, fv = ref.export()
    grp, gci, gv = M.export_ilu()
    assert np.array_equal(grp, frp) and np.array_equal(gci, fci)
    assert np.max(np.abs(gv - fv) / np.maximum(np.abs(fv), 1e-10 * np.abs(fv).max())) < 1e-10
    xoi, io, _ = orc.solve(rpi, cii, vi, bi, singular=True, prec="ilu", ilu=ref)
    bb, xg = bg.copy(), np.zeros(nl)
    info = hip.solve(octx, A, bb, xg, prec=M, singular=True)
    xo = np.empty(nl)
    xo[o["perm"]] = xoi
    assert info.converged == 1 and abs(info.iters - io.iters) <= 1, (info.iters, io.iters)
    assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) <= 1e-6
    M.close(); A.close()


def test_device_operands_and_null_mask_follow_the_callers_rows(octx):
    import torch
    dev = torch.device("cuda", 0)
    spec = tgv_spec(dim=3, n=16, mode=workload.JITTER)
    name, parts = _three_orders(spec)[2]                                   # shuffled
    kinds = [99, 12]
    parts = dict(parts)
    parts["type"] = wall_types(parts)
    P, (rp, ci, val, b) = _oracle_system(parts, spec, kinds=kinds)
    nl = parts["nlocal"]
    mask = (parts["type"][:nl] == 1).astype(np.int32)
    A, bg = _assemble(octx, parts, spec, P, kinds=kinds)
    o = A.ordering()
    M = hip.Precond(octx, A, "bjacobi-ilu0", 0)
    rpi, cii, vi, bi = oorder.permute_system(rp, ci, val, b, o["perm"])
    ref = orc.ILU(rpi, cii, vi, 0, o["block_ptr"])
    xoi, io, _ = orc.solve(rpi, cii, vi, bi, singular=True, null_mask=mask[o["perm"]], prec="ilu", ilu=ref)
    xo = np.empty(nl)
    xo[o["perm"]] = xoi
    bd = torch.from_numpy(bg.copy()).to(dev)
    xd = torch.zeros(nl, dtype=torch.float64, device=dev)
    info = hip.solve(octx, A, bd, xd, prec=M, singular=True, null_mask=mask)
    assert info.converged == 1 and abs(info.iters - io.iters) <= 1, (info.iters, io.iters)
    xg = xd.cpu().numpy()
    assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) <= 1e-6
    assert abs(xg[mask == 1].sum()) <= 1e-10 * np.abs(xg).sum()
    # device spmv / apply in the caller's numbering
    xv = torch.from_numpy(np.random.default_rng(9).standard_normal(nl)).to(dev)
    yo = sps.csr_matrix((val, ci, rp), shape=(nl, nl)) @ xv.cpu().numpy()
    assert np.max(np.abs(A.spmv(xv).cpu().numpy() - yo)) <= 1e-12 * np.abs(yo).max()
    zo = np.empty(nl)
    zo[o["perm"]] = ref.apply(xv.cpu().numpy()[o["perm"]])
    assert np.linalg.norm(M.apply(xv).cpu().numpy() - zo) / np.linalg.norm(zo) < 1e-11
    M.close(); A.close()


@pytest.mark.parametrize("mode", ["pinzero", "doublediag"])
def test_singular_modes_modify_the_callers_first_fluid_row(octx, mode):
    spec = tgv_spec(dim=3, n=12, mode=workload.JITTER)
    for name, parts in _three_orders(spec)[1:]:
        smode_o = {"pinzero": orc.PINZERO, "doublediag": orc.DOUBLEDIAG}[mode]
        smode_g = {"pinzero": hip.PINZERO, "doublediag": hip.DOUBLEDIAG}[mode]
        P, (rp, ci, val, b) = _oracle_system(parts, spec, singular=smode_o)
        A, bg = _assemble(octx, parts, spec, P, singular=smode_g)
        rp2, ci2, v2 = A.export_csr()
        assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci), name
        assert np.max(np.abs(v2 - val)) <= 1e-12 * np.abs(val).max(), name
        assert np.max(np.abs(bg - b)) <= 1e-12 * np.abs(b).max(), name
        A.close()


def test_caller_tables_and_the_librarys_numbering_do_not_mix(octx, gpu_ctx):
    spec = tgv_spec(dim=3, n=12)
    parts = workload.make_tgv(spec)
    P, _ = _oracle_system(parts, spec)
    A, _ = _assemble(octx, parts, spec, P)
    nl = parts["nlocal"]
    with pytest.raises(hip.IsphError):                                     # a table over the caller's rows means nothing here
        hip.Precond(octx, A, "bjacobi-ilu0", block_ptr=np.asarray([0, nl], dtype=np.int32) if nl <= 1024 else np.arange(0, nl + 512, 512).clip(0, nl).astype(np.int32))
    A.close()
    A2, _ = _assemble(gpu_ctx, parts, spec, P)                             # the caller's numbering: no bricks to use
    assert A2.ordering() is None
    with pytest.raises(hip.IsphError):
        hip.Precond(gpu_ctx, A2, "bjacobi-ilu0", 0)
    A2.close()


def test_periodic_box_reunites_planes_that_wrapped_around(octx):
    """isph_ctx_set_periodic_box.  A lattice whose particles have moved a little and been wrapped into [0, L), as LAMMPS
    does at every re-neighbouring: half of plane 0 sits at +eps, the other half at L - eps.  Without the box the bounding
    box grows by a spacing and the bricks hold 10.5 planes at the ends of every axis (uneven subdomains, all of which pay
    for the largest one's LDS); with it the cut goes into an empty stretch and every brick is 10 x 10 x 5 again.  Matrix
    and permutation stay exact either way."""
    spec = tgv_spec(dim=3, n=20, mode=workload.LATTICE)
    base = workload.make_tgv(spec)
    nl = base["nlocal"]
    L = 2 * np.pi
    rng = np.random.default_rng(5)
    x = base["x"][:nl] + rng.uniform(-0.02, 0.02, size=(nl, 3)) * spec.dx
    parts = workload.make_cloud(x, (L, L, L), spec.h, spec.cut, like=base)          # wraps into [0, L)
    assert parts["x"][:nl, 0].max() > L - 0.03 * spec.dx and parts["x"][:nl, 0].min() < 0.03 * spec.dx
    parts["v"] = np.zeros((parts["nall"], 3))
    parts["v"][:, 0] = np.sin(parts["x"][:, 0])
    colmap = parts["owner_index"].astype(np.int32)
    P = orc.Particles(parts, colmap, kernel=spec.kernel).precompute(corrections=False)
    rp, ci, val, b = P.poisson(spec.dt, parts["rho"], parts["v"], antisym=True)
    sizes = {}
    try:
        for label, box in (("without", None), ("with", ((0, 0, 0), (L, L, L), (1, 1, 1)))):
            if box is None:
                octx.set_periodic_box()
            else:
                octx.set_periodic_box(*box)
            A, bg = hip.assemble_poisson(octx, parts, colmap, spec.dt, parts["rho"], np.ascontiguousarray(parts["v"]), vfrac=P.vfrac)
            rp2, ci2, v2 = A.export_csr()
            assert np.array_equal(ci2, ci) and np.max(np.abs(v2 - val)) <= 1e-12 * np.abs(val).max(), label
            o = A.ordering()
            assert np.array_equal(o["perm"], oorder.order(parts["x"][:nl], o["geom"], o["faces"])), label
            fr = oorder.faces_from_histogram(parts["x"][:nl], o["geom"])
            assert all(np.array_equal(fr[a], o["faces"][a]) for a in range(3)), label
            sizes[label] = np.diff(o["block_ptr"])
            A.close()
    finally:
        octx.set_periodic_box()
    assert sizes["with"].min() == sizes["with"].max() == 500 and len(sizes["with"]) == 16
    assert sizes["without"].max() > 500


def test_correlated_cloud_splits_over_full_bricks(octx):
    """a 2-D band along the diagonal of the box: every axis sees a uniform distribution, so the quantile faces cannot thin
    the diagonal bricks out -- they hold more than 1024 particles and are cut into consecutive pieces (the table keeps
    1..1024 rows per subdomain); matrix exact, solve like the oracle's on the same table"""
    nlat, band = 128, 16
    L = 2 * np.pi
    dx = L / nlat
    ii, jj = np.meshgrid(np.arange(nlat), np.arange(nlat), indexing="ij")
    keep = (np.abs(((ii - jj + nlat // 2) % nlat) - nlat // 2) <= band).ravel()
    x = np.zeros((int(keep.sum()), 3))
    x[:, 0] = (ii.ravel()[keep] + 0.5) * dx
    x[:, 1] = (jj.ravel()[keep] + 0.5) * dx
    x[:, :2] += np.random.default_rng(3).uniform(-0.05, 0.05, size=(len(x), 2)) * dx
    spec = tgv_spec(dim=2, n=nlat, mode=workload.LATTICE)
    like = workload.make_tgv(tgv_spec(dim=2, n=16, mode=workload.LATTICE))
    parts = workload.make_cloud(x, (L, L), spec.h, spec.cut, dim=2)
    nl = parts["nlocal"]
    parts.update(spec=spec, rho=np.ones(parts["nall"]), nu=np.full(parts["nall"], 0.1), dt=spec.dt)
    parts["v"] = np.zeros((parts["nall"], 3))
    parts["v"][:, 0] = np.sin(parts["x"][:, 0]) * np.cos(parts["x"][:, 1])
    colmap = parts["owner_index"].astype(np.int32)
    P = orc.Particles(parts, colmap, kernel=spec.kernel).precompute(corrections=False)
    rp, ci, val, b = P.poisson(spec.dt, parts["rho"], parts["v"], antisym=True)
    A, bg = hip.assemble_poisson(octx, parts, colmap, spec.dt, parts["rho"], np.ascontiguousarray(parts["v"]), vfrac=P.vfrac)
    rp2, ci2, v2 = A.export_csr()
    assert np.array_equal(ci2, ci) and np.max(np.abs(v2 - val)) <= 1e-12 * np.abs(val).max()
    o = A.ordering()
    sizes = np.diff(o["block_ptr"])
    assert sizes.min() >= 1 and sizes.max() <= 1024 and o["block_ptr"][-1] == nl
    _, brick = oorder.keys(parts["x"][:nl], o["geom"], o["faces"])
    assert np.bincount(brick).max() > 1024                                 # the rule met an over-full brick ...
    assert len(sizes) > len(np.unique(brick))                              # ... and cut it
    assert np.array_equal(o["block_ptr"], oorder.block_table(parts["x"][:nl], o["geom"], o["faces"], o["perm"]))
    M = hip.Precond(octx, A, "bjacobi-ilu0", 0)
    rpi, cii, vi, bi = oorder.permute_system(rp, ci, val, b, o["perm"])
    ref = orc.ILU(rpi, cii, vi, 0, o["block_ptr"])
    xoi, io, _ = orc.solve(rpi, cii, vi, bi, singular=True, prec="ilu", ilu=ref)
    bb, xg = bg.copy(), np.zeros(nl)
    info = hip.solve(octx, A, bb, xg, prec=M, singular=True)
    assert info.converged == io.converged == 1 and abs(info.iters - io.iters) <= 1, (info.iters, io.iters)
    xo = np.empty(nl)
    xo[o["perm"]] = xoi
    assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) <= 1e-6
    M.close(); A.close()


def test_clustered_cloud_keeps_subdomains_within_bounds(octx):
    """a cloud with a dense clump: the quantile faces thin the bricks of the clump out along every axis; whatever is left
    above 1024 particles is cut into consecutive pieces; every subdomain stays within 1..1024 rows and the solve converges
    like the oracle's on the same table"""
    rng = np.random.default_rng(2)
    spec = tgv_spec(dim=3, n=16, mode=workload.JITTER)
    base = workload.make_tgv(spec)
    nl = base["nlocal"]
    L = 2 * np.pi
    x = base["x"][:nl].copy()
    sel = rng.choice(nl, size=nl // 2, replace=False)                      # half of the particles pulled into one octant
    x[sel] = (x[sel] % L) * 0.5
    parts = workload.make_cloud(x, (L, L, L), spec.h, spec.cut, like=base)
    parts["v"] = np.zeros((parts["nall"], 3))
    parts["v"][:, 0] = np.sin(parts["x"][:, 0])
    P = orc.Particles(parts, parts["owner_index"].astype(np.int32), kernel=spec.kernel).precompute(corrections=False)
    rp, ci, val, b = P.poisson(spec.dt, parts["rho"], parts["v"], antisym=True)
    A, bg = hip.assemble_poisson(octx, parts, parts["owner_index"].astype(np.int32), spec.dt, parts["rho"],
                                 np.ascontiguousarray(parts["v"]), vfrac=P.vfrac)
    rp2, ci2, v2 = A.export_csr()
    assert np.array_equal(ci2, ci) and np.max(np.abs(v2 - val)) <= 1e-12 * np.abs(val).max()
    o = A.ordering()
    sizes = np.diff(o["block_ptr"])
    assert sizes.min() >= 1 and sizes.max() <= 1024 and o["block_ptr"][-1] == nl
    assert np.array_equal(o["block_ptr"], oorder.block_table(parts["x"][:nl], o["geom"], o["faces"], o["perm"]))
    M = hip.Precond(octx, A, "bjacobi-ilu0", 0)
    rpi, cii, vi, bi = oorder.permute_system(rp, ci, val, b, o["perm"])
    ref = orc.ILU(rpi, cii, vi, 0, o["block_ptr"])
    xoi, io, _ = orc.solve(rpi, cii, vi, bi, singular=True, prec="ilu", ilu=ref)
    bb, xg = bg.copy(), np.zeros(nl)
    info = hip.solve(octx, A, bb, xg, prec=M, singular=True)
    assert info.converged == io.converged == 1 and abs(info.iters - io.iters) <= 1, (info.iters, io.iters)
    xo = np.empty(nl)
    xo[o["perm"]] = xoi
    assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) <= 1e-6
    M.close(); A.close()


def test_host_csr_with_coordinates_gets_the_librarys_numbering(octx, gpu_ctx):
    """isph_mat_create_csr_coords: the drop-in path.  The host CSR in the caller's atom order + the coordinates of its rows
    (what PrecondWrapper_ML::setCoordinates receives, precond_ml.h:63-94) give the matrix the device assembly builds from
    the particles: same permutation, same subdomain table, export in the caller's numbering bit for bit, and the solve of
    the oracle on the permuted system -- for a lexicographic and a shuffled atom order."""
    spec = tgv_spec(dim=3, n=20, mode=workload.JITTER)
    for name, parts in _three_orders(spec)[1:]:
        P, (rp, ci, val, b) = _oracle_system(parts, spec)
        nl = parts["nlocal"]
        A = hip.Matrix.from_host_csr_with_coords(gpu_ctx, rp, ci, val, parts["x"][:nl])   # any context: the call asks for it
        Ad, _ = _assemble(octx, parts, spec, P)
        o, od = A.ordering(), Ad.ordering()
        assert o is not None and np.array_equal(o["perm"], od["perm"]) and np.array_equal(o["block_ptr"], od["block_ptr"]), name
        rp2, ci2, v2 = A.export_csr()
        assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(v2, val), name
        xv = np.random.default_rng(3).standard_normal(nl)
        yo = sps.csr_matrix((val, ci, rp), shape=(nl, nl)) @ xv
        assert np.max(np.abs(A.spmv(xv) - yo)) <= 1e-12 * np.abs(yo).max(), name
        M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 0)
        rpi, cii, vi, bi = oorder.permute_system(rp, ci, val, b, o["perm"])
        ref = orc.ILU(rpi, cii, vi, 0, o["block_ptr"])
        xoi, io, _ = orc.solve(rpi, cii, vi, bi, singular=True, prec="ilu", ilu=ref)
        bb, xg = b.copy(), np.zeros(nl)
        info = hip.solve(gpu_ctx, A, bb, xg, prec=M, singular=True)
        xo = np.empty(nl)
        xo[o["perm"]] = xoi
        assert info.converged == 1 and abs(info.iters - io.iters) <= 1, (name, info.iters, io.iters)
        assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) <= 1e-6, name
        # the fused form (rows staged in the new order, set-up behind the link) is the two-step result bit for bit
        A2, M2 = hip.Matrix.from_host_csr_with_coords(gpu_ctx, rp, ci, val, parts["x"][:nl], with_bjacobi=True)
        o2 = A2.ordering()
        assert np.array_equal(o2["perm"], o["perm"]) and np.array_equal(o2["block_ptr"], o["block_ptr"]), name
        assert all(np.array_equal(a, c) for a, c in zip(A2.export_csr(), (rp, ci, val))), name
        assert np.array_equal(A2.spmv(xv), A.spmv(xv)), name
        assert all(np.array_equal(a, c) for a, c in zip(M2.export_ilu(), M.export_ilu())), name
        assert np.array_equal(M2.apply(xv), M.apply(xv)), name
        M2.close(); A2.close()
        M.close(); A.close(); Ad.close()


def test_fused_ordered_ingress_on_a_larger_matrix(gpu_ctx):
    """isph_mat_create_csr_coords_bjacobi at a size where the matrix crosses the link in several chunks and the set-up
    runs in batches behind it (40^3, shuffled atom order -- every staged row comes from somewhere else): the two-step
    result bit for bit"""
    spec = tgv_spec(dim=3, n=40, mode=workload.ADVECT)
    name, parts = _three_orders(spec)[2]
    P, (rp, ci, val, b) = _oracle_system(parts, spec)
    nl = parts["nlocal"]
    A = hip.Matrix.from_host_csr_with_coords(gpu_ctx, rp, ci, val, parts["x"][:nl])
    M = hip.Precond(gpu_ctx, A, "bjacobi-ilu0", 0)
    A2, M2 = hip.Matrix.from_host_csr_with_coords(gpu_ctx, rp, ci, val, parts["x"][:nl], with_bjacobi=True)
    xv = np.random.default_rng(3).standard_normal(nl)
    assert np.array_equal(A2.spmv(xv), A.spmv(xv))
    assert all(np.array_equal(a, c) for a, c in zip(M2.export_ilu(), M.export_ilu()))
    assert np.array_equal(M2.apply(xv), M.apply(xv))
    bb, xg = b.copy(), np.zeros(nl)
    info = hip.solve(gpu_ctx, A2, bb, xg, prec=M2, singular=True)
    bb1, xg1 = b.copy(), np.zeros(nl)
    info1 = hip.solve(gpu_ctx, A, bb1, xg1, prec=M, singular=True)
    assert info.converged == 1 and info.iters == info1.iters and np.array_equal(xg, xg1)
    for o in (M, M2, A, A2):
        o.close()


def test_cpp_mirror_with_coordinates(gpu_ctx, tmp_path):
    """PrecondWrapper_Ifpack::setCoordinates through SolverLin_Belos::solveProblem (tests/cpp/test_solver_lin.cpp, the
    adapter's three-line call of INTEGRATION.md): a shuffled atom order solves in the iterations of the library's bricks
    (the oracle on the permuted system), not in those of 512 consecutive rows of the atom order."""
    import subprocess
    from isph_amd import build
    spec = tgv_spec(dim=3, n=20, mode=workload.JITTER)
    name, parts = _three_orders(spec)[2]
    P, (rp, ci, val, b) = _oracle_system(parts, spec)
    nl = parts["nlocal"]
    exe = build.build_cpp_test()
    fin, fout, fc = tmp_path / "sys.bin", tmp_path / "x.bin", tmp_path / "coords.bin"
    with open(fin, "wb") as f:
        np.array([nl, len(val)], np.int32).tofile(f)
        rp.astype(np.int32).tofile(f); ci.astype(np.int32).tofile(f); val.tofile(f); b.tofile(f)
    np.ascontiguousarray(parts["x"][:nl].T).tofile(fc)
    r = subprocess.run([exe, str(fin), str(fout), "1", "timed", "2", "0", str(fc)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    x = np.fromfile(fout)[:nl]
    A = hip.Matrix.from_host_csr_with_coords(gpu_ctx, rp, ci, val, parts["x"][:nl])
    o = A.ordering()
    A.close()
    rpi, cii, vi, bi = oorder.permute_system(rp, ci, val, b, o["perm"])
    xoi, io, _ = orc.solve(rpi, cii, vi, bi, singular=True, prec="ilu", ilu=orc.ILU(rpi, cii, vi, 0, o["block_ptr"]))
    xo = np.empty(nl)
    xo[o["perm"]] = xoi
    its = [int(t.split(":")[1].split(",")[0]) for t in r.stdout.split('"iterations"')[1:2]]
    assert abs(its[0] - io.iters) <= 1, (its, io.iters)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6
    assert "library's bricks" in r.stdout
