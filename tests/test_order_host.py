"""CPU: the checker's side of the row numbering (oracle/order.py restates csrc/order.hpp) and the atom-order plumbing of the
generator (workload.renumber) -- no GPU, no product code path.

  * renumbering the particles (another atom order of the same cloud) permutes the oracle's system and nothing else;
  * the geometry rule on a lattice: spacing, cells, bricks of 10 x 10 x 5 / 22 x 22 with at most 20 % slack;
  * quantile faces from the histogram separate the planes of a lattice, also when the bounding box is a spacing too long
    (particles wrapped around a periodic end) -- and the declared period re-unites the wrapped plane;
  * the subdomain table: bricks in key order, over-full bricks cut into equal consecutive pieces, everything within 1..1024;
  * permute_system is a similarity transformation (P A P^T)(P x) = P (A x)."""
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sps

from isph_amd import workload
import oracle as orc
import order as oorder
from problems import tgv_spec


def _system(parts, spec):
    P = orc.Particles(parts, workload.single_rank_colmap(parts), kernel=spec.kernel).precompute(corrections=False)
    return P.poisson(spec.dt, parts["rho"], parts["v"], antisym=True)


def test_renumbering_the_atoms_permutes_the_oracle_system():
    spec = tgv_spec(dim=3, n=10, mode=workload.JITTER)
    base = workload.make_tgv(spec)
    n = base["nlocal"]
    q = np.random.default_rng(1).permutation(n)
    moved = workload.renumber(base, q)
    assert np.array_equal(moved["x"][:n], base["x"][:n][q]) and np.array_equal(moved["tag"][:n], base["tag"][:n][q])
    assert np.array_equal(np.diff(moved["neigh_ptr"]), np.diff(base["neigh_ptr"])[q])
    rp, ci, v, b = _system(base, spec)
    rp2, ci2, v2, b2 = _system(moved, spec)
    A1 = sps.csr_matrix((v, ci, rp), shape=(n, n))
    A2 = sps.csr_matrix((v2, ci2, rp2), shape=(n, n))
    assert abs(A1[q][:, q] - A2).max() == 0.0 and np.array_equal(b[q], b2)


def _geometry_with_faces(x, dim, shift=None, period=None):
    xs = x if shift is None else oorder.coords(x, SimpleNamespace(shift=shift, period=period))
    g = oorder.geometry(xs, dim)
    g.shift = list(shift) if shift is not None else [0.0] * 3
    g.period = list(period) if period is not None else [0.0] * 3
    return g, oorder.faces_from_histogram(x, g)


def test_geometry_and_faces_on_a_lattice():
    n = 40
    dx = 2 * np.pi / n
    ax = np.arange(n) * dx
    x = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    g, faces = _geometry_with_faces(x, 3)
    assert abs(g.spacing - dx) < 1e-12 and g.ncell == [n] * 3
    assert g.cells_per_brick == [10, 10, 5] and g.nbrick == [4, 4, 8]
    for a in range(3):                                     # every face lies strictly between two planes
        assert len(faces[a]) == n - 1
        assert np.all(faces[a] > ax[:-1]) and np.all(faces[a] < ax[1:])
    perm = oorder.order(x, g, faces)
    bp = oorder.block_table(x, g, faces, perm)
    assert np.all(np.diff(bp) == 500) and len(bp) - 1 == 128
    # inside a brick x runs fastest
    first = x[perm[:500]]
    assert np.allclose(first[:10, 0], ax[:10]) and np.allclose(first[:10, 1:], 0.0)


def test_wrapped_planes_and_the_declared_period():
    n = 20
    L = 2 * np.pi
    dx = L / n
    ax = np.arange(n) * dx
    x = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    x = np.mod(x + np.random.default_rng(2).uniform(-0.02, 0.02, size=x.shape) * dx, L)   # half of plane 0 lands at L - eps
    g, faces = _geometry_with_faces(x, 3)
    sizes = np.diff(oorder.block_table(x, g, faces))
    assert sizes.max() > 500 and sizes.sum() == n ** 3                  # 10.5 planes at the ends of every axis
    # the cut of isph_ctx_set_periodic_box: a point of an empty stretch that is not the box end, e.g. between planes 0 and 1
    shift, period = [0.5 * dx] * 3, [L] * 3
    g2, faces2 = _geometry_with_faces(x, 3, shift, period)
    sizes2 = np.diff(oorder.block_table(x, g2, faces2))
    assert sizes2.min() == sizes2.max() == 500 and len(sizes2) == 16


def test_over_full_bricks_are_cut_into_equal_pieces():
    rng = np.random.default_rng(3)
    x = np.zeros((5000, 3))
    x[:, :2] = rng.uniform(0, 1, size=(5000, 2))
    x[:3000, :2] = 0.5 + 1e-9 * rng.uniform(-1, 1, size=(3000, 2))      # 3000 particles in one spot (one histogram bin): no face can part them
    g, faces = _geometry_with_faces(x, 2)
    perm = oorder.order(x, g, faces)
    bp = oorder.block_table(x, g, faces, perm)
    sizes = np.diff(bp)
    assert bp[0] == 0 and bp[-1] == 5000 and sizes.min() >= 1 and sizes.max() <= oorder.BLOCK_CAP
    _, brick = oorder.keys(x, g, faces)
    big = np.bincount(brick).max()
    assert big > oorder.BLOCK_CAP
    pieces = -(-big // oorder.BLOCK_CAP)
    assert np.sum(sizes == -(-big // pieces)) >= pieces - 1            # equal pieces (the last may be shorter)
    assert np.array_equal(np.sort(perm), np.arange(5000))


def test_permute_system_is_a_similarity_transformation():
    spec = tgv_spec(dim=2, n=24, mode=workload.JITTER)
    parts = workload.make_tgv(spec)
    n = parts["nlocal"]
    rp, ci, v, b = _system(parts, spec)
    perm = np.random.default_rng(4).permutation(n).astype(np.int32)
    rpi, cii, vi, bi = oorder.permute_system(rp, ci, v, b, perm)
    assert np.all(np.diff(cii)[np.setdiff1d(np.arange(len(cii) - 1), rpi[1:-1] - 1)] > 0)   # columns ascend inside every row
    A = sps.csr_matrix((v, ci, rp), shape=(n, n))
    Ap = sps.csr_matrix((vi, cii, rpi), shape=(n, n))
    xv = np.random.default_rng(5).standard_normal(n)
    assert np.allclose(Ap @ xv[perm], (A @ xv)[perm], rtol=0, atol=1e-12 * np.abs(v).max() * 30)
    assert np.array_equal(bi, b[perm])
