"""ctypes front end of the host-side TGV particle generator (csrc/workload.cpp,
include/isph_workload.h).  Produces what the LAMMPS adapter would hand over:
atom arrays + ghosts + full neighbour list for one rank's brick."""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import build as _build

LATTICE, JITTER, ADVECT = 0, 1, 2


class _Spec(C.Structure):
    _fields_ = [("dim", C.c_int), ("ncell", C.c_int * 3), ("pgrid", C.c_int * 3), ("rank", C.c_int),
                ("brick", C.c_int * 3), ("origin", C.c_double * 3), ("h_over_dx", C.c_double),
                ("cut_over_h", C.c_double), ("skin", C.c_double), ("mode", C.c_int),
                ("jitter_amp", C.c_double), ("seed", C.c_ulonglong), ("umax", C.c_double),
                ("advect_dt", C.c_double), ("basis", C.c_int), ("colour_period", C.c_int)]


_lib = None


def _host():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build_host())
        _lib.isph_tgv_count.restype = C.c_int
        _lib.isph_tgv_fill.restype = C.c_longlong
        _lib.isph_tgv_fill64.restype = C.c_longlong
    return _lib


@dataclass
class TGVSpec:
    dim: int = 3
    ncell: tuple = (16, 16, 16)
    pgrid: tuple = (1, 1, 1)
    rank: int = 0
    brick: tuple = (8, 8, 8)
    origin: tuple = (0.0, 0.0, 0.0)
    h_over_dx: float = 1.5
    cut_over_h: float = 2.0
    skin: float = 0.0
    mode: int = ADVECT
    jitter_amp: float = 0.05
    seed: int = 42
    umax: float = 0.1
    advect_dt: float = -1.0  # <0 => 0.1*h/umax (3-D script) / 0.05*dx/umax (2-D)
    kernel: str = "wendland"
    rho: float = 1.0
    nu: float = 0.1
    basis: int = 1          # 2 = bcc (second particle at the cell centre)
    colour_period: int = 0  # > 1: multi-colour numbering inside a brick (isph_workload.h)
    extra: dict = field(default_factory=dict)

    @property
    def dx(self):
        return 2.0 * math.pi / min(self.ncell[:self.dim])     # one spacing in every direction (csrc/workload.cpp make_layout)

    @property
    def h(self):
        return self.h_over_dx * self.dx

    @property
    def cut(self):
        return self.cut_over_h * self.h

    @property
    def dt(self):
        """time step of the reference scripts: taylor-green-vortex-3d.lmp:27,
        taylor-green-vortex-2d.lmp:29"""
        if self.dim == 3:
            return 0.1 * self.h / self.umax
        return 0.05 * self.dx / self.umax

    def c_spec(self):
        s = _Spec()
        s.dim = self.dim
        nc = list(self.ncell) + [1] * (3 - len(self.ncell))
        pg = list(self.pgrid) + [1] * (3 - len(self.pgrid))
        br = list(self.brick) + [0] * (3 - len(self.brick))
        og = list(self.origin) + [0.0] * (3 - len(self.origin))
        if self.dim == 2:
            nc[2], pg[2], br[2], og[2] = 1, 1, 0, 0.0
        for a in range(3):
            s.ncell[a], s.pgrid[a], s.brick[a], s.origin[a] = nc[a], pg[a], br[a], og[a]
        s.rank = self.rank
        s.h_over_dx, s.cut_over_h, s.skin = self.h_over_dx, self.cut_over_h, self.skin
        s.mode, s.jitter_amp, s.seed, s.umax = self.mode, self.jitter_amp, self.seed, self.umax
        s.advect_dt = self.advect_dt if self.advect_dt >= 0 else self.dt
        s.basis = self.basis
        s.colour_period = self.colour_period
        return s


def make_tgv(spec: TGVSpec):
    """Returns a dict of numpy arrays for one rank: x,v [nall,3]; tag, type,
    owner_rank, owner_index [nall]; neigh_ptr [nlocal+1]; neigh_idx; plus
    scalars nlocal, nall and per-particle rho, nu."""
    lib = _host()
    cs = spec.c_spec()
    nlocal, nghost, cap = C.c_int(), C.c_int(), C.c_longlong()
    if lib.isph_tgv_count(C.byref(cs), C.byref(nlocal), C.byref(nghost), C.byref(cap)) != 0:
        raise ValueError("invalid TGV spec")
    nlocal, nall = nlocal.value, nlocal.value + nghost.value
    x = np.zeros((nall, 3))
    v = np.zeros((nall, 3))
    tag = np.zeros(nall, dtype=np.int32)
    orank = np.zeros(nall, dtype=np.int32)
    oidx = np.zeros(nall, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    # count pass first (64-bit offsets), then a list of exactly that size: the stencil bound `cap` is 3-4x the list
    nptr = np.zeros(nlocal + 1, dtype=np.int64)
    nn = lib.isph_tgv_fill64(C.byref(cs), p(x), p(v), p(tag), p(orank), p(oidx), p(nptr), None)
    if nn < 0:
        raise RuntimeError("isph_tgv_fill64 failed")
    nidx = np.zeros(max(int(nn), 1), dtype=np.int32)
    nn = lib.isph_tgv_fill64(C.byref(cs), p(x), p(v), p(tag), p(orank), p(oidx), p(nptr), p(nidx))
    nidx = nidx[:nn]
    if nn < 2 ** 31 - 1:
        nptr = nptr.astype(np.int32)     # LAMMPS-like 32-bit offsets whenever they fit; int64 -> neigh_ptr64
    return dict(spec=spec, dim=spec.dim, nlocal=nlocal, nall=nall, x=x, v=v, tag=tag,
                type=np.ones(nall, dtype=np.int32), owner_rank=orank, owner_index=oidx,
                neigh_ptr=nptr, neigh_idx=nidx,
                rho=np.full(nall, spec.rho), nu=np.full(nall, spec.nu),
                h=spec.h, cut=spec.cut, dt=spec.dt)


def make_cloud(x_owned, box, h, cut, dim=3, like=None):
    """Ghost atoms + full neighbour list of a general particle cloud in a periodic box (isph_cloud_build): the dict
    make_tgv returns, for owned positions x_owned [nlocal, 3] (wrapped into the box here).  `like`: a make_tgv dict whose
    spec / per-particle constants (rho, nu, type of the owned particles) are carried over; tags are 1 + the owned index."""
    lib = _host()
    lib.isph_cloud_build.restype = C.c_longlong
    lib.isph_cloud_build.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L = np.ascontiguousarray(list(box) + [1.0] * (3 - len(box)), dtype=np.float64)
    x = np.ascontiguousarray(x_owned, dtype=np.float64).copy()
    for a in range(dim):
        x[:, a] = np.mod(x[:, a], L[a])
        x[x[:, a] >= L[a], a] = 0.0                  # mod can return L for tiny negatives
    nl = x.shape[0]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    ng = lib.isph_cloud_build(dim, nl, p(x), p(L), float(cut), None, None, None, None)
    if ng < 0:
        raise ValueError("isph_cloud_build: bad arguments (box shorter than two cuts?)")
    nall = nl + int(ng)
    xa = np.zeros((nall, 3))
    own = np.zeros(nall, dtype=np.int32)
    nptr = np.zeros(nl + 1, dtype=np.int64)
    nn = lib.isph_cloud_build(dim, nl, p(x), p(L), float(cut), p(xa), p(own), p(nptr), None)
    nidx = np.zeros(max(int(nn), 1), dtype=np.int32)
    nn = lib.isph_cloud_build(dim, nl, p(x), p(L), float(cut), p(xa), p(own), p(nptr), p(nidx))
    nidx = nidx[:nn]
    if nn < 2 ** 31 - 1:
        nptr = nptr.astype(np.int32)
    spec = like["spec"] if like is not None else None
    out = dict(spec=spec, dim=dim, nlocal=nl, nall=nall, x=xa, v=np.zeros((nall, 3)), tag=(own + 1).astype(np.int32),
               type=np.ones(nall, dtype=np.int32), owner_rank=np.zeros(nall, dtype=np.int32), owner_index=own,
               neigh_ptr=nptr, neigh_idx=nidx, h=float(h), cut=float(cut))
    if like is not None:
        for k in ("rho", "nu"):
            out[k] = np.ascontiguousarray(like[k][:nl][own])
        out["type"] = np.ascontiguousarray(like["type"][:nl][own])
        out["dt"] = like.get("dt")
    return out


def single_rank_colmap(parts):
    """Matrix column of every particle on one rank: ghosts are periodic images,
    their column is the owner's local id (Epetra LID of the shared tag)."""
    assert np.all(parts["owner_rank"] == parts["spec"].rank)
    return parts["owner_index"].astype(np.int32).copy()


FLUID_KIND, SOLID_KIND = 99, 12     # PairISPH::ParticleKind (pair_isph.h:113-138)


def make_cavity(nfluid, wall=6, dim=3, pgrid=(1, 1, 1), rank=0, brick=(8, 8, 8), umax=5.0, nu=0.1, rho=1.0, lid_inset=4,
                jitter=0.0, seed=42):
    """Closed lid-driven cavity of sph-script/lid-driven-cavity-{2d,3d}.m + .lmp (BASELINE configs[3]): a simple-cubic
    lattice of (nfluid + 2 wall)^dim sites in a periodic box; the inner nfluid^dim sites are fluid (type 1), the
    `wall` layers around them solid (type 2), the part of the +y wall above the fluid minus `lid_inset` columns at
    the x/z rims is the moving lid (type 3, velocity (umax,0,0)) (.m: `type(X.^2 < (box_half_x-4*dx)^2 & ... & Y > 0 &
    tmp == 2) = 3`, .lmp: `velocity surface set ${Umax} 0 0`).  The script reads a data file written by the .m
    script; this generator produces the same lattice, labels and velocities directly, in units where the periodic box
    is [0, 2 pi).  Besides the arrays of make_tgv the dict carries
      kinds   [FLUID, SOLID, SOLID]  ("type:1 fluid:moving, type:2/3 solid:fixed", lid-driven-cavity.xml)
      normal  [nall][3] unit normals pointing into the fluid on the solid particles within the cut of the fluid
              (the reference gets them from computeNormals, out of scope: they are an input here)
      dt      0.1 h / Umax (.lmp: tstep)
    wall must cover the cut (wall * dx >= cut) so the fluid never sees its periodic image."""
    ncell = nfluid + 2 * wall
    nc = (ncell,) * dim
    spec = TGVSpec(dim=dim, ncell=nc, pgrid=pgrid, rank=rank, brick=brick[:dim], origin=(0.0,) * dim,
                   mode=JITTER if jitter > 0 else LATTICE, jitter_amp=jitter, seed=seed, umax=umax, nu=nu, rho=rho)
    assert wall * spec.dx >= spec.cut - 1e-12, "wall thinner than the kernel support"
    p = make_tgv(spec)
    tag0 = p["tag"].astype(np.int64) - 1
    idx = np.stack([tag0 % ncell, (tag0 // ncell) % ncell, tag0 // (ncell * ncell)], axis=1)[:, :dim]   # lattice site
    inside = np.all((idx >= wall) & (idx < wall + nfluid), axis=1)
    typ = np.where(inside, 1, 2).astype(np.int32)
    rim = np.ones(len(typ), dtype=bool)
    for a in range(dim):
        if a != 1:
            rim &= (idx[:, a] >= wall + lid_inset) & (idx[:, a] < wall + nfluid - lid_inset)
    typ[(idx[:, 1] >= wall + nfluid) & rim & (typ == 2)] = 3
    v = np.zeros((p["nall"], 3))
    v[typ == 3, 0] = umax
    nrm = np.zeros((p["nall"], 3))
    reach = int(np.ceil(spec.cut / spec.dx))
    for a in range(dim):
        nrm[idx[:, a] < wall, a] = 1.0
        nrm[idx[:, a] >= wall + nfluid, a] = -1.0
    near = np.all((idx >= wall - reach) & (idx < wall + nfluid + reach), axis=1) & ~inside
    nrm[~near] = 0.0
    ln = np.linalg.norm(nrm, axis=1)
    nrm[ln > 0] /= ln[ln > 0, None]
    p.update(type=typ, v=v, normal=nrm, kinds=[FLUID_KIND, SOLID_KIND, SOLID_KIND], dt=0.1 * spec.h / umax,
             nfluid=nfluid, wall=wall)
    return p


def make_porous_cylinder(nc, wall=5, nbeads=12, rbead_cells=3.0, seed=7, brick=(8, 8, 8), pgrid=(1, 1, 1), rank=0,
                         jitter=0.0, rho=997.561, nu=8.9087e-07, umax=0.04, bead_pack=None):
    """Pore-scale flow through a bead pack in a cylinder (BASELINE configs[4]; sph-script/pore-scale-flow-3d.lmp +
    compute_isph_cylinder_porous.cpp:195-224): `lattice bcc ${dx}` (2 particles per cell, 749 entries per matrix row
    with the script's Quintic kernel, cut = 3 h = 4.5 dx), cylinder along y (periodic), labels
        type 4  outside the cylinder radius (solid wall)          is_coords_in_cylinder, compute_isph_cylinder_porous.h:65-71
        type 3  inside a bead (solid; beads only in the middle half of the length, :45-63)
        type 2  buffer fluid (a slab of the length near the inlet, :73-75)
        type 1  fluid
    with "type:1/2 fluid, type:3/4 solid:fixed" (pore-scale-flow.xml).  The reference creates atoms only inside the
    radius r + 4 dx of a non-periodic box; here the lattice fills a periodic cube of nc cells per side whose corners
    are additional type-4 wall (wall >= cut/dx cells thick so the fluid never sees its periodic image), the bead
    centres are drawn from a seeded generator instead of the script's data file, and beads are solid throughout
    (inner radius 0: no deleted particles).

    bead_pack: the script's own bead pack (tests/golden/pore_scale_flow_bead_centeroids_3d.npz: the 6864 centres of
    pore-scale-flow-bead-centeroids-3d.dat with the script's r, half length, bead radius and buffer slab).  The box then
    has the script's aspect ratio (length / diameter = 1.634, periodic along y like `boundary f p f`), the beads whose
    centre lies in the middle half of the length are solid (:49-52), lengths scaled so that the cylinder radius is
    (nc/2 - wall) cells.  Beads stay solid throughout here as well: the reference deletes the particles deeper than
    one cut inside a bead, which at <= 8 cells per bead radius is nothing or a handful."""
    ncy = nc
    if bead_pack is not None:
        r_phys, hl_phys = float(bead_pack["r"]), float(bead_pack["half_length"])
        ncy = int(round(hl_phys / r_phys * (nc - 2 * wall)))
        ncy += ncy % 2                                       # whole bricks of the lattice along y
    spec = TGVSpec(dim=3, ncell=(nc, ncy, nc), pgrid=pgrid, rank=rank, brick=brick, origin=(0.0, 0.0, 0.0),
                   mode=JITTER if jitter > 0 else LATTICE, jitter_amp=jitter, seed=seed, umax=umax, nu=nu, rho=rho,
                   basis=2, kernel="quintic", cut_over_h=3.0)
    assert wall * spec.dx >= spec.cut - 1e-12, "wall thinner than the kernel support"
    p = make_tgv(spec)
    L, dx = 2.0 * math.pi, spec.dx
    x = p["x"]
    xw = np.mod(x, L)                                        # labels are periodic: images get their owner's label
    c = 0.5 * L
    R = (0.5 * nc - wall) * dx
    r2 = (xw[:, 0] - c) ** 2 + (xw[:, 2] - c) ** 2
    typ = np.ones(p["nall"], dtype=np.int32)
    part = np.zeros(p["nall"], dtype=np.int32)
    if bead_pack is not None:
        from scipy.spatial import cKDTree
        Ly = ncy * dx
        xw[:, 1] = np.mod(x[:, 1], Ly)
        scale = R / r_phys
        b0, b1 = (float(t) * scale for t in bead_pack["buffer"])
        typ[(xw[:, 1] > b0) & (xw[:, 1] < b1)] = 2           # buffer slab behind the inlet (:73-75)
        ctr = np.asarray(bead_pack["centres"], dtype=np.float64)
        mid = np.abs(ctr[:, 1]) < 0.5 * hl_phys               # beadlo < y < beadhi (pore-scale-flow-3d.lmp:122-123)
        ctr = ctr[mid]
        bc = np.stack([c + scale * ctr[:, 0], scale * (ctr[:, 1] + hl_phys), c + scale * ctr[:, 2]], axis=1)
        rb = float(bead_pack["rbead"]) * scale
        dist, idx = cKDTree(bc).query(xw, distance_upper_bound=rb)
        hit = np.isfinite(dist)
        typ[hit] = 3
        part[hit] = idx[hit] + 1
        nbeads = 0
    else:
        typ[(xw[:, 1] > 0.05 * L) & (xw[:, 1] < 0.12 * L)] = 2   # buffer slab
        rb = rbead_cells * dx
    rng = np.random.default_rng(seed)
    for k in range(nbeads):
        rad = (R - rb) * math.sqrt(rng.random())
        ang = 2.0 * math.pi * rng.random()
        bc = np.array([c + rad * math.cos(ang), (0.25 + 0.5 * rng.random()) * L, c + rad * math.sin(ang)])
        d2 = np.sum((xw - bc) ** 2, axis=1)
        hit = (d2 < rb * rb) & (part == 0)
        typ[hit] = 3
        part[hit] = k + 1
    typ[r2 >= R * R] = 4
    part[r2 >= R * R] = -1
    v = np.zeros((p["nall"], 3))
    p.update(type=typ, v=v, part=part, kinds=[FLUID_KIND, FLUID_KIND, SOLID_KIND, SOLID_KIND],
             g=np.array([0.0, 1.06, 0.0]), dt=0.1 * spec.h / umax, radius=R)
    return p


def renumber(parts, order):
    """The same particles in another atom order, as LAMMPS would hold them after sorting / migration: new owned particle
    k is the old owned particle order[k]; ghosts keep their slots (their owner_index follows).  Every per-particle array
    of the dict, the neighbour list (rows moved, entries renamed) and the tags follow.  Single rank only (the images'
    owners are renumbered with the same permutation)."""
    n, nall = int(parts["nlocal"]), int(parts["nall"])
    order = np.asarray(order, dtype=np.int64)
    assert order.shape == (n,) and np.array_equal(np.sort(order), np.arange(n))
    assert np.all(parts["owner_rank"] == parts["owner_rank"][0]), "renumber: single rank only"
    inv = np.empty(n, dtype=np.int64)
    inv[order] = np.arange(n)
    full = np.r_[order, np.arange(n, nall)]                   # gather index over all particles
    idmap = np.r_[inv, np.arange(n, nall)]                    # old particle index -> new
    out = dict(parts)
    for k, a in parts.items():
        if isinstance(a, np.ndarray) and a.ndim >= 1 and a.shape[0] == nall and k not in ("neigh_idx",):
            out[k] = np.ascontiguousarray(a[full])
    out["owner_index"] = idmap[parts["owner_index"][full]].astype(parts["owner_index"].dtype)
    nptr = parts["neigh_ptr"].astype(np.int64)
    lens = (nptr[1:] - nptr[:-1])[order]
    nptr2 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=nptr2[1:])
    src = np.repeat(nptr[:-1][order] - nptr2[:-1], lens) + np.arange(int(nptr2[-1]), dtype=np.int64)
    out["neigh_idx"] = idmap[parts["neigh_idx"][src]].astype(np.int32)
    out["neigh_ptr"] = nptr2.astype(parts["neigh_ptr"].dtype)
    return out
