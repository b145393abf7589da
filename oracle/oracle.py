"""ctypes wrapper of the CPU ORACLE (oracle/isph_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
KERNELS = {"wendland": 0, "quintic": 1, "cubic": 2}
FLUID, SOLID, ALL = 99, 12, 127
NOT_SINGULAR, NULLSPACE, PINZERO, DOUBLEDIAG = 0, 1, 2, 3


class _Particles(C.Structure):
    _fields_ = [("dim", C.c_int), ("nlocal", C.c_int), ("nall", C.c_int), ("ntypes", C.c_int),
                ("kernel", C.c_int),
                ("x", C.c_void_p), ("type", C.c_void_p), ("kind", C.c_void_p), ("h", C.c_void_p),
                ("cutsq", C.c_void_p), ("neigh_ptr", C.c_void_p), ("neigh_idx", C.c_void_p),
                ("colmap", C.c_void_p), ("owner", C.c_void_p), ("vfrac", C.c_void_p),
                ("Gc", C.c_void_p), ("Lc", C.c_void_p), ("pnd", C.c_void_p),
                ("morris_safe_coeff", C.c_double)]


class SolverParams(C.Structure):
    """Defaults = SolverLin_Belos::setParameters, solver_lin_belos.h:224-264."""
    _fields_ = [("solver_type", C.c_int), ("flexible", C.c_int), ("num_blocks", C.c_int),
                ("max_iters", C.c_int), ("max_restarts", C.c_int), ("tol", C.c_double),
                ("ortho", C.c_int), ("verbose", C.c_int)]

    def __init__(self, solver_type=0, flexible=1, num_blocks=50, max_iters=500, max_restarts=15,
                 tol=1e-8, ortho=0, verbose=0):
        super().__init__(solver_type, flexible, num_blocks, max_iters, max_restarts, tol, ortho, verbose)


class SolveInfo(C.Structure):
    _fields_ = [("converged", C.c_int), ("iters", C.c_int), ("restarts", C.c_int),
                ("rel_res_implicit", C.c_double), ("rel_res_explicit", C.c_double),
                ("setup_seconds", C.c_double), ("solve_seconds", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_oracle_build", os.path.join(_HERE, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        path = mod.build_oracle()
        L = C.CDLL(path)
        L.orc_kernel_val.restype = C.c_double
        L.orc_kernel_dval.restype = C.c_double
        L.orc_kernel_val.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_kernel_dval.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_ilu_create.restype = C.c_void_p
        L.orc_ilu_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_ilu_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ilu_nnz.argtypes = [C.c_void_p]
        L.orc_ilu_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ilu_destroy.argtypes = [C.c_void_p]
        L.orc_schwarz_create.restype = C.c_void_p
        L.orc_schwarz_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_schwarz_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_schwarz_nloc.argtypes = [C.c_void_p]
        L.orc_schwarz_nnz.argtypes = [C.c_void_p]
        L.orc_schwarz_export.argtypes = [C.c_void_p] * 6
        L.orc_schwarz_destroy.argtypes = [C.c_void_p]
        L.orc_amg_create.restype = C.c_void_p
        L.orc_amg_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_double, C.c_int, C.c_int, C.c_double]
        L.orc_amg_create_ex.restype = C.c_void_p
        L.orc_amg_create_ex.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.orc_amg_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_amg_set_smoother.argtypes = [C.c_void_p, C.c_int]
        L.orc_amg_levels.argtypes = [C.c_void_p]
        L.orc_amg_level_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_amg_export.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_amg_export_aggregates.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_amg_destroy.argtypes = [C.c_void_p]
        L.orc_laplacian_matrix.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_divergence.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]
        L.orc_gradient.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]
        L.orc_compute_shift.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_apply_shift.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int]
        L.orc_laplacian_apply.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p,
                                          C.c_int, C.c_int, C.c_void_p]
        L.orc_poisson.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
        L.orc_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_solute_transport.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_applied_potential.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]
        L.orc_helmholtz.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_void_p]
        L.orc_block_helmholtz.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_spmv.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_solve_block.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_forward_comm.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        # GPU boxes expose 256 hardware threads but give a job a 16-CPU share: cap the team size
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        L.orc_set_num_threads(int(os.environ.get("ISPH_ORACLE_THREADS", min(16, ncpu))))
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Particles:
    """Holds the arrays alive and exposes the C view."""

    def __init__(self, parts, colmap, kernel="wendland", kinds=None, h=None, cut=None, pnd=None,
                 morris_safe_coeff=0.0):
        self.dim = int(parts["dim"])
        self.nlocal, self.nall = int(parts["nlocal"]), int(parts["nall"])
        self.x = _f64(parts["x"])
        self.type = _i32(parts["type"])
        ntypes = int(self.type.max())
        self.kind = _i32([0] + list(kinds if kinds is not None else [FLUID] * ntypes))
        hh = parts["h"] if h is None else h
        cc = parts["cut"] if cut is None else cut
        self.h = _f64(np.full((ntypes + 1, ntypes + 1), hh))
        self.cutsq = _f64(np.full((ntypes + 1, ntypes + 1), cc * cc))
        self.neigh_ptr, self.neigh_idx = _i32(parts["neigh_ptr"]), _i32(parts["neigh_idx"])
        self.colmap = _i32(colmap)
        own = np.where(parts["owner_rank"] == parts["spec"].rank, parts["owner_index"], -1) \
            if "owner_rank" in parts else np.arange(self.nall)
        self.owner = _i32(own)
        dL = self.dim * (self.dim + 1) // 2
        self.vfrac = np.zeros(self.nall)
        self.Gc = np.zeros((self.nall, self.dim * self.dim))
        self.Lc = np.zeros((self.nall, dL))
        self.pnd = None if pnd is None else _f64(pnd)
        self.c = _Particles(self.dim, self.nlocal, self.nall, ntypes, KERNELS[kernel],
                            _p(self.x), _p(self.type), _p(self.kind), _p(self.h), _p(self.cutsq),
                            _p(self.neigh_ptr), _p(self.neigh_idx), _p(self.colmap), _p(self.owner),
                            _p(self.vfrac), _p(self.Gc), _p(self.Lc), _p(self.pnd), morris_safe_coeff)

    def ref(self):
        return C.byref(self.c)

    def precompute(self, corrections=True):
        """computePre: volumes, then (Symmetric family only) G_i and L_i.
        ref: pair_isph_corrected.cpp:302-369."""
        L = lib()
        L.orc_compute_volumes(self.ref())
        if corrections:
            L.orc_compute_gradient_correction(self.ref())
            L.orc_compute_laplacian_correction(self.ref())
        return self

    def compute_pnd(self):
        """particle number density of the MorrisHolmes mirror ([nall], ghosts filled): functor_normal.h:57-133"""
        pnd = np.zeros(self.nall)
        lib().orc_compute_pnd(self.ref(), _p(pnd))
        return pnd

    def graph(self):
        cap = int(self.neigh_ptr[-1]) + self.nlocal
        rowptr = np.zeros(self.nlocal + 1, dtype=np.int32)
        colidx = np.zeros(cap, dtype=np.int32)
        nnz = lib().orc_graph(self.ref(), _p(rowptr), _p(colidx), cap)
        assert nnz >= 0
        return rowptr, colidx[:nnz].copy()

    def laplacian_matrix(self, rowptr, colidx, antisym, alpha, material=None, filt=(FLUID, ALL), morris=0):
        val = np.zeros(len(colidx))
        m = None if material is None else _f64(material)
        bad = lib().orc_laplacian_matrix(self.ref(), int(antisym), float(alpha), _p(m), filt[0], filt[1],
                                         morris, _p(rowptr), _p(colidx), _p(val))
        assert bad == 0, "entries outside graph"
        return val

    def divergence(self, f, antisym, alpha=1.0, filt=None, morris=0):
        f = _f64(f)
        div = np.zeros(self.nlocal)
        use = filt is not None
        fi, fj = filt if use else (ALL, ALL)
        lib().orc_divergence(self.ref(), int(antisym), _p(f), float(alpha), int(use), fi, fj, morris, _p(div))
        return div

    def gradient(self, f, antisym, alpha=1.0, filt=None):
        f = _f64(f)
        g = np.zeros((self.nlocal, 3))
        use = filt is not None
        fi, fj = filt if use else (ALL, ALL)
        lib().orc_gradient(self.ref(), int(antisym), _p(f), float(alpha), int(use), fi, fj, _p(g))
        return g

    def compute_shift(self, alpha, shiftcut, nonfluidweight):
        dr = np.zeros((self.nlocal, 3))
        lib().orc_compute_shift(self.ref(), float(alpha), float(shiftcut), float(nonfluidweight), _p(dr))
        return dr

    def apply_shift(self, antisym, dr, v, p, fixed=None, sequential=False):
        """returns shifted copies (x, v, p); `fixed` is per type (index 0 unused)."""
        x, v, p = self.x.copy(), _f64(v).copy(), _f64(p).copy()
        fx = None if fixed is None else _i32(fixed)
        lib().orc_apply_shift(self.ref(), int(antisym), _p(fx), _p(_f64(dr)), _p(x), _p(v), _p(p), int(sequential))
        return x, v, p

    def laplacian_apply(self, f, antisym, alpha, material=None, filt=(FLUID, ALL)):
        f = _f64(f)
        ncomp = 1 if f.ndim == 1 else f.shape[1]
        out = np.zeros((self.nlocal, ncomp))
        m = None if material is None else _f64(material)
        lib().orc_laplacian_apply(self.ref(), int(antisym), _p(f), ncomp, float(alpha), _p(m), filt[0], filt[1],
                                  _p(out))
        return out

    def forward_comm(self, arr):
        arr = _f64(arr)
        ncomp = 1 if arr.ndim == 1 else arr.shape[1]
        lib().orc_forward_comm(self.ref(), _p(arr), ncomp)
        return arr

    def poisson(self, dt, rho, vstar, antisym=True, singular=NULLSPACE, normal=None, morris=0, rank0=True,
                graph=None, solid_normal_diag=1.0):
        rowptr, colidx = graph if graph is not None else self.graph()
        val = np.zeros(len(colidx))
        b = np.zeros(self.nlocal)
        work = np.zeros(self.nall)
        rho, vstar = _f64(rho), _f64(vstar)
        nrm = None if normal is None else _f64(normal)
        rc = lib().orc_poisson(self.ref(), int(antisym), morris, float(dt), _p(rho), _p(vstar), _p(nrm),
                               float(solid_normal_diag), singular, int(rank0), _p(rowptr), _p(colidx), _p(val), _p(b), _p(work))
        assert rc == 0, "orc_poisson rc=%d" % rc
        return rowptr, colidx, val, b


def _helmholtz(self, dt, theta, nu, rho, p, f, g, vall, antisym=True, incremental=True, graph=None, morris=0):
    """computeHelmholtz: returns (rowptr, colidx, val, b[nlocal, dim] column-major as [dim][nlocal])."""
    rowptr, colidx = graph if graph is not None else self.graph()
    val = np.zeros(len(colidx))
    n = self.nlocal
    vall = _f64(vall)
    b = np.ascontiguousarray(vall[:n, :self.dim].T.copy())      # column-major [lda x dim], holds v^n
    work = np.zeros(self.nall)
    nu, rho, p, f, g = _f64(nu), _f64(rho), _f64(p), _f64(f), _f64(g)
    rc = lib().orc_helmholtz(self.ref(), int(antisym), int(morris), float(dt), float(theta), _p(nu), _p(rho), _p(p), _p(f), _p(g),
                             int(incremental), _p(vall), _p(rowptr), _p(colidx), _p(val), _p(b), n, _p(work))
    assert rc == 0, "orc_helmholtz rc=%d" % rc
    return rowptr, colidx, val, b


Particles.helmholtz = _helmholtz


FILTER_MATCH = 0x1000   # or-ed into filt[0]: FilterMatchBinary instead of FilterBinary (isph_oracle.h)
BUFFER_DIRICHLET, BUFFER_NEUMANN = 32, 64


def _solute_transport(self, dt, theta, dcoeff, conc, antisym=True, graph=None):
    """computeSoluteTransportSpecies: returns (rowptr, colidx, val, b[nlocal])."""
    rowptr, colidx = graph if graph is not None else self.graph()
    val = np.zeros(len(colidx))
    b = np.zeros(self.nlocal)
    rc = lib().orc_solute_transport(self.ref(), int(antisym), float(dt), float(theta), float(dcoeff), _p(_f64(conc)),
                                    _p(rowptr), _p(colidx), _p(val), _p(b))
    assert rc == 0, "orc_solute_transport rc=%d" % rc
    return rowptr, colidx, val, b


def _applied_potential(self, sigma, phi, antisym=True, graph=None):
    """computeAppliedElectricPotential: returns (rowptr, colidx, val, b[nlocal])."""
    rowptr, colidx = graph if graph is not None else self.graph()
    val = np.zeros(len(colidx))
    b = np.zeros(self.nlocal)
    sg = None if sigma is None else _f64(sigma)
    rc = lib().orc_applied_potential(self.ref(), int(antisym), _p(sg), _p(_f64(phi)), _p(rowptr), _p(colidx), _p(val), _p(b))
    assert rc == 0, "orc_applied_potential rc=%d" % rc
    return rowptr, colidx, val, b


Particles.solute_transport = _solute_transport
Particles.applied_potential = _applied_potential


def _block_helmholtz(self, dt, theta, beta, nu, rho, p, f, g, vall, normal=None, antisym=True, incremental=True,
                     graph=None, morris=0):
    """computeBlockHelmholtz: returns (rowptr, colidx, vals[dim*dim, nnz], b[dim][nlocal]); block (ib,jb) = vals[ib*dim+jb]."""
    rowptr, colidx = graph if graph is not None else self.graph()
    d2 = self.dim * self.dim
    vals = np.zeros((d2, len(colidx)))
    n = self.nlocal
    vall = _f64(vall)
    b = np.ascontiguousarray(vall[:n, :self.dim].T.copy())
    nu, rho, p, f, g = _f64(nu), _f64(rho), _f64(p), _f64(f), _f64(g)
    nrm = None if normal is None else _f64(normal)
    rc = lib().orc_block_helmholtz(self.ref(), int(antisym), int(morris), float(dt), float(theta), float(beta), _p(nu),
                                   _p(rho), _p(p), _p(f), _p(g), int(incremental), _p(nrm), _p(vall), _p(rowptr),
                                   _p(colidx), _p(vals), _p(b), n)
    assert rc == 0, "orc_block_helmholtz rc=%d" % rc
    return rowptr, colidx, vals, b


Particles.block_helmholtz = _block_helmholtz


def kernel_val(kernel, dim, r, h):
    return lib().orc_kernel_val(KERNELS[kernel], dim, float(r), float(h))


def kernel_dval(kernel, dim, r, h):
    return lib().orc_kernel_dval(KERNELS[kernel], dim, float(r), float(h))


def spmv(rowptr, colidx, val, x):
    y = np.zeros(len(rowptr) - 1)
    x = _f64(x)
    lib().orc_spmv(len(rowptr) - 1, _p(rowptr), _p(colidx), _p(val), _p(x), _p(y))
    return y


class ILU:
    """Block-Jacobi ILU(k) == Ifpack AdditiveSchwarz<ILU>, overlap 0."""

    def __init__(self, rowptr, colidx, val, level_of_fill=0, block_ptr=None):
        self.n = len(rowptr) - 1
        self._keep = (_i32(rowptr), _i32(colidx), _f64(val))
        bp = None if block_ptr is None else _i32(block_ptr)
        nb = 0 if bp is None else len(bp) - 1
        self.h = lib().orc_ilu_create(self.n, _p(self._keep[0]), _p(self._keep[1]), _p(self._keep[2]),
                                      level_of_fill, nb, _p(bp))

    def apply(self, r):
        r = _f64(r)
        z = np.zeros(self.n)
        lib().orc_ilu_apply(self.h, _p(r), _p(z))
        return z

    def export(self):
        nnz = lib().orc_ilu_nnz(self.h)
        rp = np.zeros(self.n + 1, dtype=np.int32)
        ci = np.zeros(nnz, dtype=np.int32)
        v = np.zeros(nnz)
        lib().orc_ilu_export(self.h, _p(rp), _p(ci), _p(v))
        return rp, ci, v

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().orc_ilu_destroy(self.h)
            except TypeError:        # interpreter shutdown: the module globals are already gone
                pass
            self.h = None


class Schwarz:
    """Ifpack_AdditiveSchwarz<ILU(k)> with overlap (isph_schwarz_oracle.c).  own_ptr = consecutive owned row ranges
    (None: one subdomain = the whole matrix, the reference on one MPI rank); combine "add" (reference) | "zero"."""

    def __init__(self, rowptr, colidx, val, level_of_fill=0, own_ptr=None, overlap=0, combine="add"):
        self.n = len(rowptr) - 1
        self._keep = (_i32(rowptr), _i32(colidx), _f64(val))
        op = _i32([0, self.n] if own_ptr is None else own_ptr)
        self.nsub = len(op) - 1
        self.h = lib().orc_schwarz_create(self.n, _p(self._keep[0]), _p(self._keep[1]), _p(self._keep[2]),
                                          int(level_of_fill), self.nsub, _p(op), int(overlap),
                                          {"add": 0, "zero": 1}[combine])

    def apply(self, r):
        r = _f64(r)
        z = np.zeros(self.n)
        lib().orc_schwarz_apply(self.h, _p(r), _p(z))
        return z

    def export(self):
        """(rows[nloc], loc_ptr[nsub+1], factor CSR in local numbering)"""
        nloc, nnz = lib().orc_schwarz_nloc(self.h), lib().orc_schwarz_nnz(self.h)
        rows = np.zeros(nloc, dtype=np.int32)
        lp = np.zeros(self.nsub + 1, dtype=np.int32)
        rp = np.zeros(nloc + 1, dtype=np.int32)
        ci = np.zeros(nnz, dtype=np.int32)
        v = np.zeros(nnz)
        lib().orc_schwarz_export(self.h, _p(rows), _p(lp), _p(rp), _p(ci), _p(v))
        return rows, lp, rp, ci, v

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().orc_schwarz_destroy(self.h)
            except TypeError:
                pass
            self.h = None


class AMG:
    """Smoothed-aggregation AMG standing in for PrecondWrapper_ML (isph_amg_oracle.c)."""

    def __init__(self, rowptr, colidx, val, nullvec=None, max_levels=5, coarse_max=128, omega=4.0 / 3.0, block=512,
                 sweeps=1, theta=0.0, aggregation="mis2", whole_sgs=False, smoother=0):
        """aggregation "mis2" = the device algorithm, "ml" = ML's sequential Uncoupled sweep; whole_sgs = Gauss-Seidel
        over the whole level (ML on one rank) instead of block-local"""
        self.n = len(rowptr) - 1
        self._keep = (_i32(rowptr), _i32(colidx), _f64(val), None if nullvec is None else _f64(nullvec))
        self.h = lib().orc_amg_create_ex(self.n, _p(self._keep[0]), _p(self._keep[1]), _p(self._keep[2]),
                                         _p(self._keep[3]), max_levels, coarse_max, omega, block, sweeps, theta,
                                         {"mis2": 0, "ml": 1}[aggregation], int(whole_sgs))
        if smoother:                      # 1 = "ML Gauss-Seidel", efficient symmetric (ml.xml of the benchmark protocol)
            lib().orc_amg_set_smoother(self.h, int(smoother))

    @property
    def levels(self):
        return lib().orc_amg_levels(self.h)

    def level_info(self, l):
        a = np.zeros(3, dtype=np.int32)
        lib().orc_amg_level_info(self.h, l, _p(a))
        return dict(rows=int(a[0]), nnz=int(a[1]), nnz_p=int(a[2]))

    def export(self, l, what="A"):
        i = self.level_info(l)
        nnz = i["nnz"] if what == "A" else i["nnz_p"]
        rp = np.zeros(i["rows"] + 1, dtype=np.int32)
        ci = np.zeros(nnz, dtype=np.int32)
        v = np.zeros(nnz)
        lib().orc_amg_export(self.h, l, 0 if what == "A" else 1, _p(rp), _p(ci), _p(v))
        return rp, ci, v

    def aggregates(self, l):
        a = np.zeros(self.level_info(l)["rows"], dtype=np.int32)
        lib().orc_amg_export_aggregates(self.h, l, _p(a))
        return a

    def apply(self, r):
        r = _f64(r)
        z = np.zeros(self.n)
        lib().orc_amg_apply(self.h, _p(r), _p(z))
        return z

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().orc_amg_destroy(self.h)
            except TypeError:        # interpreter shutdown
                pass
            self.h = None


def solve(rowptr, colidx, val, b, x0=None, singular=False, null_mask=None, prec="none", ilu=None,
          params=None, amg=None, schwarz=None):
    """SolverLin_Belos::solveProblem restatement.  Returns (x, info, b_projected)."""
    n = len(rowptr) - 1
    rowptr, colidx, val = _i32(rowptr), _i32(colidx), _f64(val)
    b = _f64(b).copy()
    x = np.zeros(n) if x0 is None else _f64(x0).copy()
    prm = params or SolverParams()
    info = SolveInfo()
    mask = None if null_mask is None else _i32(null_mask)
    ptype = {"none": 0, "jacobi": 1, "ilu": 2, "amg": 3, "schwarz": 4}[prec]
    obj = amg.h if prec == "amg" else (schwarz.h if prec == "schwarz" else (ilu.h if ilu is not None else None))
    lib().orc_solve(n, _p(rowptr), _p(colidx), _p(val), _p(b), _p(x), int(singular), _p(mask), ptype,
                    obj, C.byref(prm), C.byref(info))
    return x, info, b


def solve_block(rowptr, colidx, val, b, dim, x0=None, prec="none", ilu=None, amg=None, params=None):
    """SolverLin_Belos::solveBlockProblem restatement: the blocked operator as one CSR over [x_0;..;x_{dim-1}],
    the preconditioner object (one block's size) applied to every component."""
    n = len(rowptr) - 1
    rowptr, colidx, val = _i32(rowptr), _i32(colidx), _f64(val)
    b = _f64(b).copy()
    x = np.zeros(n) if x0 is None else _f64(x0).copy()
    prm = params or SolverParams()
    info = SolveInfo()
    ptype = {"none": 0, "ilu": 2, "amg": 3}[prec]
    obj = amg.h if prec == "amg" else (ilu.h if prec == "ilu" else None)
    lib().orc_solve_block(n, dim, _p(rowptr), _p(colidx), _p(val), _p(b), _p(x), ptype, obj, C.byref(prm),
                          C.byref(info))
    return x, info


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
