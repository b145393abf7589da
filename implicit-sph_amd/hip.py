"""ctypes binding of libisph_hip.so (include/isph_hip.h) -- the product path.

Nothing in here touches the oracle or any CPU fallback: if the HIP library is
missing or no GPU is usable, calls raise.  Arrays may be numpy (host) or torch
CUDA tensors (device pointers are passed straight through).
"""
import ctypes as C
import sys
import os

import numpy as np

from . import build as _build

UID_BYTES = 128
NOT_SINGULAR, NULLSPACE, PINZERO, DOUBLEDIAG = 0, 1, 2, 3
KERNELS = {"wendland": 0, "quintic": 1, "cubic": 2}

EXPORTS = [
    "isph_ctx_create", "isph_comm_unique_id", "isph_ctx_create_dist", "isph_ctx_create_hostcomm", "isph_ctx_sync", "isph_ctx_destroy", "isph_pool_trim", "isph_pool_set_cap", "isph_set_exact_stream_threshold", "isph_pool_cached_bytes", "isph_pool_info", "isph_halo_create", "isph_halo_forward", "isph_halo_destroy", "isph_prec_create_overlap",
    "isph_last_error", "isph_mat_create_csr", "isph_mat_create_csr_bjacobi", "isph_mat_create_csr_blocks", "isph_mat_create_csr_coords", "isph_mat_create_csr_coords_bjacobi", "isph_ingress_info", "isph_mat_set_halo", "isph_mat_info", "isph_mat_export_csr", "isph_mat_export_rows",
    "isph_mat_destroy", "isph_spmv", "isph_spmv_time", "isph_prec_create", "isph_prec_create_blocks", "isph_prec_create_blocks_fill", "isph_prec_apply",
    "isph_prec_export_ilu", "isph_prec_nnz", "isph_prec_info", "isph_prec_destroy", "isph_solver_params_default", "isph_solve",
    "isph_ctx_set_profile", "isph_ctx_hold_neighbours", "isph_ctx_profile_read", "isph_ctx_set_ordering", "isph_ctx_set_periodic_box", "isph_mat_ordering_info", "isph_mat_ordering", "isph_mat_ordering_faces", "isph_ctx_halo_profile_read", "isph_ctx_comm_info", "isph_device_identity", "isph_assemble_poisson", "isph_assemble_helmholtz", "isph_assemble_solute_transport", "isph_assemble_applied_potential", "isph_compute_volumes", "isph_compute_pnd", "isph_compute_corrections", "isph_gradient", "isph_divergence", "isph_correct_velocity_pressure",
    "isph_advance_begin", "isph_advance_end", "isph_compute_shift", "isph_apply_shift", "isph_shift_particles",
    "isph_solve_block", "isph_assemble_block_helmholtz", "isph_amg_params_default", "isph_prec_create_amg", "isph_prec_amg_levels", "isph_prec_amg_info",
    "isph_prec_amg_export", "isph_prec_amg_aggregates",
    "isph_schwarz_params_default", "isph_prec_create_schwarz", "isph_prec_schwarz_info", "isph_prec_schwarz_timing", "isph_prec_schwarz_export",
]


class AmgParams(C.Structure):
    """Mirror of isph_amg_params == the keys PrecondWrapper_ML::setParameters sets (precond_ml.h:44-55)."""
    _fields_ = [("max_levels", C.c_int), ("coarse_max", C.c_int), ("omega", C.c_double), ("block", C.c_int),
                ("sweeps", C.c_int), ("theta", C.c_double), ("smoother", C.c_int)]

    def __init__(self, **kw):
        super().__init__()
        lib().isph_amg_params_default(C.byref(self))
        for k, v in kw.items():
            setattr(self, k, v)


class SolverParams(C.Structure):
    """Mirror of isph_solver_params == SolverLin_Belos::setParameters keys
    (ref: solver_lin_belos.h:224-264)."""
    _fields_ = [("solver_type", C.c_int), ("flexible", C.c_int), ("num_blocks", C.c_int),
                ("max_iters", C.c_int), ("max_restarts", C.c_int), ("tol", C.c_double),
                ("ortho", C.c_int), ("verbose", C.c_int), ("num_recycled", C.c_int)]

    def __init__(self, solver_type=0, flexible=1, num_blocks=50, max_iters=500, max_restarts=15, tol=1e-8,
                 ortho=0, verbose=0, num_recycled=50):
        """solver_type 0 "Block GMRES", 1 "Block CG", 2 "Recycling GMRES" (GCRO-DR(num_blocks, num_recycled))"""
        super().__init__(solver_type, flexible, num_blocks, max_iters, max_restarts, tol, ortho, verbose, num_recycled)


class SolveInfo(C.Structure):
    _fields_ = [("converged", C.c_int), ("iters", C.c_int), ("restarts", C.c_int),
                ("rel_res_implicit", C.c_double), ("rel_res_explicit", C.c_double),
                ("prec_setup_ms", C.c_double), ("solve_ms", C.c_double), ("spmv_ms", C.c_double),
                ("spmv_calls", C.c_int), ("reorth", C.c_int)]


class OrderGeometry(C.Structure):
    """isph_order_geometry: what the library's brick sort of the owned particles was made with (isph_mat_ordering_info)."""
    _fields_ = [("dim", C.c_int), ("lo", C.c_double * 3), ("inv_bin", C.c_double * 3), ("nbins", C.c_int * 3), ("ncell", C.c_int * 3),
                ("cells_per_brick", C.c_int * 3), ("nbrick", C.c_int * 3), ("shift", C.c_double * 3), ("period", C.c_double * 3)]


ORDERINGS = {"caller": 0, "bricks": 1}
# Row numbering of the contexts this binding creates when the caller does not say: None = the library's default
# (ISPH_ORDER_BRICKS: the assembly sorts the owned particles into bricks itself).  tests/conftest.py sets "caller" for the
# suites that compare internals (ILU factors, AMG aggregates) row by row with the oracle in the generator's numbering.
DEFAULT_ORDERING = None


class _Particles(C.Structure):
    _fields_ = [("dim", C.c_int), ("nlocal", C.c_int), ("nall", C.c_int), ("ntypes", C.c_int),
                ("kernel", C.c_int), ("x", C.c_void_p), ("type", C.c_void_p), ("kind", C.c_void_p),
                ("h", C.c_void_p), ("cutsq", C.c_void_p), ("neigh_ptr", C.c_void_p), ("neigh_idx", C.c_void_p),
                ("colmap", C.c_void_p), ("vfrac", C.c_void_p), ("Gc", C.c_void_p), ("Lc", C.c_void_p),
                ("morris_holmes", C.c_int), ("pnd", C.c_void_p), ("morris_safe_coeff", C.c_double),
                ("normal", C.c_void_p), ("solid_normal_diag", C.c_double), ("neigh_ptr64", C.c_void_p)]


_lib = None


def lib_path():
    return _build.HIP_LIB


def lib():
    """Loads libisph_hip.so; raises (never falls back) if it is not there."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError("libisph_hip.so is missing (%s): run __graft_entry__.build(); "
                               "there is no CPU fallback" % path)
        # When torch is in the process, load it first so both bind to ONE HIP/RCCL runtime
        # (torch ships its own libamdhip64/librccl; two runtimes in one process do not share the device).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(path)
        L.isph_last_error.restype = C.c_char_p
        L.isph_prec_nnz.restype = C.c_longlong
        L.isph_prec_nnz.argtypes = [C.c_void_p]
        L.isph_ctx_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.isph_ctx_create_dist.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_void_p]
        L.isph_ctx_create_hostcomm.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.isph_ctx_sync.argtypes = [C.c_void_p]
        L.isph_ctx_destroy.argtypes = [C.c_void_p]
        L.isph_pool_trim.argtypes = []
        L.isph_halo_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.isph_halo_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.isph_halo_destroy.argtypes = [C.c_void_p]
        L.isph_halo_destroy.restype = None
        L.isph_prec_create_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.isph_pool_set_cap.argtypes = [C.c_longlong]
        L.isph_set_exact_stream_threshold.argtypes = [C.c_longlong]
        L.isph_ingress_info.argtypes = [C.c_void_p, C.c_void_p]
        L.isph_pool_info.argtypes = [C.c_void_p, C.c_int]
        L.isph_pool_cached_bytes.argtypes = []
        L.isph_pool_cached_bytes.restype = C.c_longlong
        L.isph_ctx_set_profile.argtypes = [C.c_void_p, C.c_int]
        L.isph_ctx_hold_neighbours.argtypes = [C.c_void_p, C.c_int]
        L.isph_ctx_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_ctx_halo_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_ctx_comm_info.argtypes = [C.c_void_p, C.c_void_p]
        L.isph_device_identity.argtypes = [C.c_int, C.c_char_p]
        L.isph_ctx_set_ordering.argtypes = [C.c_void_p, C.c_int]
        L.isph_ctx_set_periodic_box.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_mat_ordering_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_mat_ordering.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_mat_ordering_faces.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.isph_mat_create_csr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p]
        L.isph_mat_create_csr_bjacobi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.c_void_p, C.c_void_p]
        L.isph_mat_create_csr_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_mat_create_csr_coords.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_mat_create_csr_coords_bjacobi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_mat_set_halo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.isph_mat_info.argtypes = [C.c_void_p, C.c_void_p]
        L.isph_mat_export_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_mat_export_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        L.isph_mat_destroy.argtypes = [C.c_void_p]
        L.isph_spmv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.isph_spmv_time.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.isph_prec_create.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
        L.isph_prec_create_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.isph_prec_create_blocks_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.isph_prec_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.isph_prec_export_ilu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_prec_destroy.argtypes = [C.c_void_p]
        L.isph_prec_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.isph_assemble_poisson.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.isph_compute_volumes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.isph_compute_pnd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.isph_compute_corrections.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        for fn in (L.isph_gradient, L.isph_divergence):
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int,
                           C.c_void_p, C.c_int]
        L.isph_correct_velocity_pressure.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.isph_advance_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int]
        L.isph_advance_end.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int]
        L.isph_assemble_block_helmholtz.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_int]
        L.isph_solve_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_int]
        L.isph_prec_create_schwarz.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_prec_schwarz_info.argtypes = [C.c_void_p, C.c_void_p]
        L.isph_prec_schwarz_export.argtypes = [C.c_void_p] * 7
        L.isph_prec_schwarz_timing.argtypes = [C.c_void_p, C.c_void_p]
        L.isph_amg_params_default.argtypes = [C.c_void_p]
        L.isph_prec_create_amg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.isph_prec_amg_levels.argtypes = [C.c_void_p]
        L.isph_prec_amg_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.isph_prec_amg_export.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.isph_prec_amg_aggregates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.isph_compute_shift.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int]
        L.isph_apply_shift.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int]
        L.isph_shift_particles.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int]
        L.isph_assemble_helmholtz.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.isph_assemble_solute_transport.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                                     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.isph_assemble_applied_potential.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                      C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


class IsphError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise IsphError(lib().isph_last_error().decode() or "isph call failed")


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a):
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def _on_device(*arrs):
    flags = {bool(_is_torch(a) and a.is_cuda) for a in arrs if a is not None}
    if len(flags) != 1:
        raise ValueError("mix of host and device arrays")
    return int(flags.pop())


def _i32(a):
    if _is_torch(a):
        assert str(a.dtype) == "torch.int32", "int32 tensor expected, got %s" % a.dtype
        return a
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    if _is_torch(a):
        assert str(a.dtype) == "torch.float64", "float64 tensor expected, got %s" % a.dtype
        return a
    return np.ascontiguousarray(a, dtype=np.float64)


def device_identity(device=0):
    """isph_device_identity: the PCI bus id of HIP device `device` as this process sees it"""
    buf = C.create_string_buffer(64)
    _check(lib().isph_device_identity(int(device), buf))
    return buf.value.decode()


def pool_trim():
    """Return the device buffers the library keeps for the next set-up to the driver (isph_pool_trim)."""
    _check(lib().isph_pool_trim())


def ingress_info(ctx):
    """isph_ingress_info: milestones of the context's last host-side matrix ingress (ms), bytes that crossed the link."""
    a = (C.c_double * 8)()
    _check(lib().isph_ingress_info(ctx.h, a))
    return dict(staged_ms=a[0], queued_ms=a[1], copied_ms=a[2], device_done_ms=a[3], end_ms=a[4], waited_for_staging_ms=a[5],
                link_bytes=int(a[6]), threads=int(a[7]))


def pool_set_cap(nbytes):
    """Limit the library's cache of freed device blocks (isph_pool_set_cap; <= 0 restores the default of 80 % of the
    memory that was free at the first release).  torch's caching allocator does not see this cache: a process that
    lets torch allocate large tensors next to the library either caps it here or calls pool_trim() when
    torch.cuda.OutOfMemoryError is raised and retries."""
    _check(lib().isph_pool_set_cap(int(nbytes)))


def set_exact_stream_threshold(nbytes):
    """isph_set_exact_stream_threshold: ILU / Gauss-Seidel streams whose capacity-rule reservation exceeds nbytes are sized
    by a counting pass instead (default 4 GiB; 0 = always exact; < 0 = default)."""
    _check(lib().isph_set_exact_stream_threshold(int(nbytes)))


def pool_info(reset_peak=False):
    """isph_pool_info: dict(cached, live, peak_live, cap) in bytes."""
    a = (C.c_longlong * 4)()
    _check(lib().isph_pool_info(a, int(reset_peak)))
    return dict(cached=a[0], live=a[1], peak_live=a[2], cap=a[3])


def pool_cached_bytes():
    return int(lib().isph_pool_cached_bytes())


class HostTransport(C.Structure):
    """isph_host_transport: the two callbacks of a host-staged communicator (isph_ctx_create_hostcomm)."""
    _fields_ = [("user", C.c_void_p), ("exchange", C.c_void_p), ("allreduce", C.c_void_p)]


class Context:
    """isph_ctx: device + stream (+ RCCL communicator when nranks > 1, or the host-staged transport `transport`, a
    HostTransport whose callbacks outlive the context, for ranks that share a device)."""

    def __init__(self, device=0, stream=None, rank=0, nranks=1, uid=None, transport=None, ordering=None):
        """ordering: "bricks" (the library numbers the matrix rows itself, isph_ctx_set_ordering) | "caller" | None =
        DEFAULT_ORDERING, else the library's default (bricks)"""
        self.h = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        if stream is None and "torch" in sys.modules:
            # default to the stream torch is issuing on, so zero-fills / temporaries of the caller and the
            # library's kernels are stream-ordered
            import torch
            if torch.cuda.is_available():
                stream = torch.cuda.current_stream(device).cuda_stream
        # handle 0 is the legacy null stream: the library then creates its own stream with hipStreamDefault,
        # i.e. one that is implicitly ordered against null-stream work (isph_capi.hip ctx_create_common)
        sp = C.c_void_p(stream) if stream else None
        if transport is not None:
            _check(lib().isph_ctx_create_hostcomm(device, sp, rank, nranks, C.byref(transport), C.byref(self.h)))
        elif nranks > 1 or uid is not None:
            _check(lib().isph_ctx_create_dist(device, sp, rank, nranks, uid, C.byref(self.h)))
        else:
            _check(lib().isph_ctx_create(device, sp, C.byref(self.h)))
        ordering = DEFAULT_ORDERING if ordering is None else ordering
        self.ordering = "bricks" if ordering is None else ordering
        if ordering is not None:
            self.set_ordering(ordering)

    def set_ordering(self, mode):
        """isph_ctx_set_ordering: "bricks" | "caller" for the matrices assembled from now on"""
        _check(lib().isph_ctx_set_ordering(self.h, ORDERINGS[mode]))
        self.ordering = mode

    def set_periodic_box(self, lo=None, hi=None, periodic=None):
        """isph_ctx_set_periodic_box: the caller's periodic box for the brick sort (None forgets it)"""
        if lo is None:
            _check(lib().isph_ctx_set_periodic_box(self.h, None, None, None))
            return
        a = (C.c_double * 3)(*[float(v) for v in (list(lo) + [0.0] * 3)[:3]])
        b = (C.c_double * 3)(*[float(v) for v in (list(hi) + [0.0] * 3)[:3]])
        p = (C.c_int * 3)(*[int(v) for v in (list(periodic) + [0] * 3)[:3]])
        _check(lib().isph_ctx_set_periodic_box(self.h, a, b, p))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(UID_BYTES)
        _check(lib().isph_comm_unique_id(buf))
        return buf.raw

    def sync(self):
        _check(lib().isph_ctx_sync(self.h))

    def set_profile(self, on):
        _check(lib().isph_ctx_set_profile(self.h, int(on)))

    def hold_neighbours(self, on):
        """isph_ctx_hold_neighbours: while held, the operator calls share one layout of the (unchanged) neighbour list"""
        _check(lib().isph_ctx_hold_neighbours(self.h, int(on)))

    PROFILE_CLASSES = ("spmv", "prec_apply", "multi_dot", "multi_axpy_dot", "multi_axpy_norm", "ilu_extract", "ilu_schedule",
                       "ilu_factor")

    def profile_read(self):
        """isph_ctx_profile_read: {class: (milliseconds, launches)} since the collection started; starts the next one"""
        ms, calls = (C.c_double * 8)(), (C.c_int * 8)()
        _check(lib().isph_ctx_profile_read(self.h, ms, calls))
        return {k: (ms[i], calls[i]) for i, k in enumerate(self.PROFILE_CLASSES)}

    def halo_profile_read(self):
        """isph_ctx_halo_profile_read: dict(products, exchange_ms, interior_ms, exposed_ms) summed since the collection started"""
        ms, calls = (C.c_double * 3)(), C.c_int()
        _check(lib().isph_ctx_halo_profile_read(self.h, ms, C.byref(calls)))
        return dict(products=calls.value, exchange_ms=ms[0], interior_ms=ms[1], exposed_ms=ms[2])

    def comm_info(self):
        """isph_ctx_comm_info: dict(transport "none" | "rccl" | "host", ranks, rank, device) -- for RCCL from the communicator"""
        a = (C.c_longlong * 4)()
        _check(lib().isph_ctx_comm_info(self.h, a))
        return dict(transport=("none", "rccl", "host")[int(a[0])], ranks=int(a[1]), rank=int(a[2]), device=int(a[3]))

    def close(self):
        if self.h:
            lib().isph_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HaloForward:
    """Stand-alone halo plan: forward comm of per-atom fields over the context's RCCL communicator
    (isph_halo_create / isph_halo_forward; LAMMPS' comm->forward_comm_pair for PairISPH, pair_isph.cpp:1924-2110)."""

    def __init__(self, ctx, nlocal, peers, send_ptr, send_idx, recv_ptr):
        self.ctx, self.nlocal = ctx, int(nlocal)
        peers, send_ptr, send_idx, recv_ptr = (np.ascontiguousarray(a, dtype=np.int32) for a in (peers, send_ptr, send_idx, recv_ptr))
        self.nrecv = int(recv_ptr[-1]) if len(recv_ptr) else 0
        self.h = C.c_void_p()
        _check(lib().isph_halo_create(ctx.h, self.nlocal, len(peers), _ptr(peers), _ptr(send_ptr), _ptr(send_idx), _ptr(recv_ptr),
                                      C.byref(self.h)))

    def forward(self, x, ncomp=1):
        """x: [nlocal] or [nlocal, ncomp] (numpy on the host or torch on the device) -> ghost values [nrecv(, ncomp)]
        in ghost-column order, on the same side as x."""
        dev = _on_device(x)
        x = _f64(x)
        assert x.shape[0] >= self.nlocal and (x.ndim == 1) == (ncomp == 1) and (x.ndim == 1 or x.shape[1] == ncomp)
        shape = (self.nrecv,) if ncomp == 1 else (self.nrecv, ncomp)
        if dev:
            import torch
            g = torch.empty(shape, dtype=torch.float64, device=x.device)
        else:
            g = np.empty(shape, dtype=np.float64)
        _check(lib().isph_halo_forward(self.ctx.h, self.h, _ptr(x), _ptr(g), int(ncomp), 1 if dev else 0))
        return g

    def close(self):
        if self.h:
            lib().isph_halo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Matrix:
    """isph_mat: sliced-ELL device matrix (SolverLin::setMatrix ingress)."""

    def __init__(self, ctx, handle=None):
        self.ctx = ctx
        self.h = handle if handle is not None else C.c_void_p()

    @classmethod
    def from_csr(cls, ctx, rowptr, colidx, val, ncol=None):
        rowptr, colidx, val = _i32(rowptr), _i32(colidx), _f64(val)
        nrow = int(rowptr.shape[0]) - 1
        m = cls(ctx)
        _check(lib().isph_mat_create_csr(ctx.h, nrow, nrow if ncol is None else ncol, _ptr(rowptr), _ptr(colidx),
                                         _ptr(val), _on_device(rowptr, colidx, val), C.byref(m.h)))
        return m

    @classmethod
    def from_host_csr_with_coords(cls, ctx, rowptr, colidx, val, coords, dim=3, ncol=None, with_bjacobi=False):
        """isph_mat_create_csr_coords: host CSR in the caller's atom order + the coordinates of its rows ([nrow, >= dim]) ->
        a matrix in the library's own row numbering (the drop-in path with PrecondWrapper_Ifpack::setCoordinates).
        with_bjacobi: isph_mat_create_csr_coords_bjacobi, returns (Matrix, Precond) with the ILU(0) set-up of the library's
        bricks fused with the ingress"""
        rowptr, colidx, val = _i32(rowptr), _i32(colidx), _f64(val)
        nrow = int(rowptr.shape[0]) - 1
        xs = [np.ascontiguousarray(coords[:nrow, a], dtype=np.float64) for a in range(dim)]
        m = cls(ctx)
        if with_bjacobi:
            M = Precond.__new__(Precond)
            M.ctx, M.n, M.h = ctx, nrow, C.c_void_p()
            _check(lib().isph_mat_create_csr_coords_bjacobi(ctx.h, nrow, nrow if ncol is None else ncol, _ptr(rowptr), _ptr(colidx),
                                                            _ptr(val), int(dim), _ptr(xs[0]), _ptr(xs[1]),
                                                            _ptr(xs[2]) if dim == 3 else None, C.byref(m.h), C.byref(M.h)))
            return m, M
        _check(lib().isph_mat_create_csr_coords(ctx.h, nrow, nrow if ncol is None else ncol, _ptr(rowptr), _ptr(colidx), _ptr(val),
                                                int(dim), _ptr(xs[0]), _ptr(xs[1]), _ptr(xs[2]) if dim == 3 else None, C.byref(m.h)))
        return m

    @classmethod
    def from_host_csr_with_bjacobi(cls, ctx, rowptr, colidx, val, block_size=512, ncol=None, block_ptr=None):
        """isph_mat_create_csr_bjacobi: host CSR ingress fused with the block-Jacobi ILU(0) set-up (the drop-in path of
        SolverLin_Belos::solveProblem with PrecondWrapper_Ifpack).  block_ptr: the caller's subdomains
        (isph_mat_create_csr_blocks).  Returns (Matrix, Precond)."""
        rowptr, colidx, val = _i32(rowptr), _i32(colidx), _f64(val)
        if _is_torch(rowptr) or _is_torch(colidx) or _is_torch(val):
            raise IsphError("isph_mat_create_csr_bjacobi takes host arrays")
        nrow = int(rowptr.shape[0]) - 1
        m = cls(ctx)
        M = Precond.__new__(Precond)
        M.ctx, M.n, M.h = ctx, nrow, C.c_void_p()
        if block_ptr is not None:
            bp = np.ascontiguousarray(block_ptr, dtype=np.int32)
            _check(lib().isph_mat_create_csr_blocks(ctx.h, nrow, nrow if ncol is None else ncol, _ptr(rowptr), _ptr(colidx),
                                                    _ptr(val), len(bp) - 1, _ptr(bp), C.byref(m.h), C.byref(M.h)))
            return m, M
        _check(lib().isph_mat_create_csr_bjacobi(ctx.h, nrow, nrow if ncol is None else ncol, _ptr(rowptr), _ptr(colidx),
                                                 _ptr(val), block_size, C.byref(m.h), C.byref(M.h)))
        return m, M

    def set_halo(self, peers, send_ptr, send_idx, recv_ptr):
        peers, send_ptr, send_idx, recv_ptr = map(lambda a: np.ascontiguousarray(a, dtype=np.int32),
                                                  (peers, send_ptr, send_idx, recv_ptr))
        _check(lib().isph_mat_set_halo(self.ctx.h, self.h, len(peers), _ptr(peers), _ptr(send_ptr), _ptr(send_idx),
                                       _ptr(recv_ptr)))
        self._halo = True

    def info(self):
        a = (C.c_longlong * 6)()
        _check(lib().isph_mat_info(self.h, a))
        return dict(nrow=a[0], ncol=a[1], nnz=a[2], nslices=a[3], stored=a[4], sell_bytes=a[5])

    def ordering(self):
        """None for a matrix in the caller's row numbering; else dict(perm [nrow]: the caller's row held by internal row r,
        block_ptr: the library's subdomains over internal rows, geom: OrderGeometry, faces: the cell faces of the three axes)
        -- isph_mat_ordering(_info), isph_mat_ordering_faces."""
        a = (C.c_longlong * 3)()
        g = OrderGeometry()
        _check(lib().isph_mat_ordering_info(self.h, a, C.byref(g)))
        if not a[0]:
            return None
        perm = np.zeros(int(a[1]), dtype=np.int32)
        bp = np.zeros(int(a[2]) + 1, dtype=np.int32)
        _check(lib().isph_mat_ordering(self.ctx.h, self.h, _ptr(perm), _ptr(bp)))
        faces = []
        for ax in range(3):
            f = np.zeros(max(int(g.ncell[ax]) - 1, 0))
            if ax < g.dim and len(f):
                _check(lib().isph_mat_ordering_faces(self.h, ax, _ptr(f)))
            faces.append(f)
        return dict(perm=perm, block_ptr=bp, geom=g, faces=faces)

    def subdomains(self):
        """the library's subdomain table alone (no copy of the permutation); None for a matrix in the caller's numbering"""
        a = (C.c_longlong * 3)()
        _check(lib().isph_mat_ordering_info(self.h, a, None))
        if not a[0]:
            return None
        bp = np.zeros(int(a[2]) + 1, dtype=np.int32)
        _check(lib().isph_mat_ordering(self.ctx.h, self.h, None, _ptr(bp)))
        return bp

    def export_csr(self):
        i = self.info()
        rp = np.zeros(i["nrow"] + 1, dtype=np.int32)
        ci = np.zeros(i["nnz"], dtype=np.int32)
        v = np.zeros(i["nnz"])
        _check(lib().isph_mat_export_csr(self.ctx.h, self.h, _ptr(rp), _ptr(ci), _ptr(v)))
        return rp, ci, v

    def export_rows(self, row_begin, nrows, capacity=None):
        """isph_mat_export_rows: rows [row_begin, row_begin + nrows) as (rowptr int64 relative, colidx, val)."""
        cap = int(capacity if capacity is not None else nrows * 4096)
        rp = np.zeros(nrows + 1, dtype=np.int64)
        ci = np.zeros(cap, dtype=np.int32)
        v = np.zeros(cap)
        _check(lib().isph_mat_export_rows(self.ctx.h, self.h, int(row_begin), int(nrows), _ptr(rp), _ptr(ci), _ptr(v), cap))
        return rp, ci[:rp[-1]], v[:rp[-1]]

    def sampled_product(self, x, nsamples=64, rows_per_sample=64, seed=0):
        """(A x) on `nsamples` runs of `rows_per_sample` consecutive rows, computed ON THE HOST from exported rows: an
        operator application that shares nothing with the SpMV kernels.  Returns (row indices, values)."""
        n = self.info()["nrow"]
        xh = x.detach().cpu().numpy() if _is_torch(x) else np.asarray(x)
        rng = np.random.default_rng(seed)
        starts = np.unique(np.minimum(rng.integers(0, max(n - rows_per_sample, 0) + 1, size=nsamples), max(n - rows_per_sample, 0)))
        rows, vals, mags = [], [], []
        for s0 in starts:
            m = min(rows_per_sample, n - int(s0))
            rp, ci, v = self.export_rows(int(s0), m)
            prod = v * xh[ci]
            vals.append(np.add.reduceat(prod, rp[:-1]) * (np.diff(rp) > 0) if len(prod) else np.zeros(m))
            mags.append(np.add.reduceat(np.abs(prod), rp[:-1]) * (np.diff(rp) > 0) if len(prod) else np.zeros(m))
            rows.append(np.arange(int(s0), int(s0) + m))
        self.last_sample_magnitude = np.concatenate(mags)     # sum_j |a_ij x_j| per sampled row: the scale of its round-off
        return np.concatenate(rows), np.concatenate(vals)

    def spmv(self, x, y=None):
        x = _f64(x)
        inf = self.info()
        # ghost columns without a halo plan: the caller supplies all ncol entries; otherwise the nrow owned ones
        _need(x, inf["ncol"] if (inf["ncol"] > inf["nrow"] and not getattr(self, "_halo", False)) else inf["nrow"], "x")
        _need(y, inf["nrow"], "y [nrow]")
        if y is None:
            n = self.info()["nrow"]
            if _is_torch(x):
                import torch
                y = torch.empty(n, dtype=torch.float64, device=x.device)
            else:
                y = np.zeros(n)
        _check(lib().isph_spmv(self.ctx.h, self.h, _ptr(x), _ptr(y), _on_device(x, y)))
        return y

    def spmv_time(self, x, y, reps=20, variant=0):
        ms = C.c_double()
        _check(lib().isph_spmv_time(self.ctx.h, self.h, _ptr(x), _ptr(y), reps, int(variant), C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            lib().isph_mat_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Precond:
    """isph_prec == PrecondWrapper_Ifpack::create() result."""

    def __init__(self, ctx, A, kind="bjacobi-ilu0", block_size=512, block_ptr=None):
        """block_ptr: the caller's subdomains (isph_prec_create_blocks; kind must be "bjacobi-ilu0"): ascending row
        offsets from 0 to nrow, at most 1024 rows per subdomain.  block_size 0: the matrix' own subdomains (the bricks of
        the library's row numbering)"""
        self.ctx, self.n = ctx, A.info()["nrow"]
        self.h = C.c_void_p()
        if block_ptr is not None:
            assert kind.startswith("bjacobi-ilu") and kind[11:].isdigit()
            bp = np.ascontiguousarray(block_ptr, dtype=np.int32)
            _check(lib().isph_prec_create_blocks_fill(ctx.h, A.h, len(bp) - 1, _ptr(bp), int(kind[11:]), C.byref(self.h)))
        else:
            _check(lib().isph_prec_create(ctx.h, A.h, kind.encode(), block_size, C.byref(self.h)))

    def apply(self, r, z=None):
        r = _f64(r)
        _need(r, self.n, "r [n]"); _need(z, self.n, "z [n]")
        if z is None:
            if _is_torch(r):
                import torch
                z = torch.empty_like(r)
            else:
                z = np.zeros(self.n)
        _check(lib().isph_prec_apply(self.ctx.h, self.h, _ptr(r), _ptr(z), _on_device(r, z)))
        return z

    def info(self):
        a = (C.c_longlong * 4)()
        _check(lib().isph_prec_info(self.ctx.h, self.h, a))
        return dict(factor_nnz=a[0], stream_chunks=a[1], stream_capacity=a[2], nblocks=a[3])

    def export_ilu(self):
        nnz = lib().isph_prec_nnz(self.h)
        rp = np.zeros(self.n + 1, dtype=np.int32)
        ci = np.zeros(nnz, dtype=np.int32)
        v = np.zeros(nnz)
        _check(lib().isph_prec_export_ilu(self.ctx.h, self.h, _ptr(rp), _ptr(ci), _ptr(v)))
        return rp, ci, v

    def close(self):
        if self.h:
            lib().isph_prec_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SchwarzParams(C.Structure):
    _fields_ = [("level_of_fill", C.c_int), ("overlap", C.c_int), ("combine", C.c_int), ("block_size", C.c_int),
                ("level_launches", C.c_int)]


class PrecondSchwarz(Precond):
    """isph_prec_create_schwarz: Ifpack_AdditiveSchwarz<ILU(k)> (precond_ifpack.h:28-75).  block_size 0 = one
    subdomain = the whole local matrix (the reference on one rank); combine "add" (reference) | "zero"."""

    def __init__(self, ctx, A, level_of_fill=1, overlap=1, combine="add", block_size=0, level_launches=False):
        self.ctx, self.h, self.n = ctx, C.c_void_p(), A.info()["nrow"]
        prm = SchwarzParams(int(level_of_fill), int(overlap), {"add": 0, "zero": 1}[combine], int(block_size),
                            int(bool(level_launches)))
        _check(lib().isph_prec_create_schwarz(ctx.h, A.h, C.byref(prm), C.byref(self.h)))

    def schwarz_info(self):
        a = (C.c_longlong * 7)()
        _check(lib().isph_prec_schwarz_info(self.h, a))
        return dict(nloc=a[0], nnz=a[1], nsub=a[2], levels_l=a[3], levels_u=a[4], maxrow=a[5], persistent=a[6])

    def create_timing(self):
        """ms of the create call by stage (isph_prec_schwarz_timing)."""
        a = (C.c_double * 6)()
        _check(lib().isph_prec_schwarz_timing(self.h, a))
        return dict(to_host=a[0], local_matrices=a[1], pattern=a[2], levels=a[3], upload=a[4], factor=a[5])

    def export(self):
        i = self.schwarz_info()
        rows = np.zeros(i["nloc"], dtype=np.int32)
        lp = np.zeros(i["nsub"] + 1, dtype=np.int32)
        rp = np.zeros(i["nloc"] + 1, dtype=np.int64)
        ci = np.zeros(i["nnz"], dtype=np.int32)
        v = np.zeros(i["nnz"])
        _check(lib().isph_prec_schwarz_export(self.ctx.h, self.h, _ptr(rows), _ptr(lp), _ptr(rp), _ptr(ci), _ptr(v)))
        return rows, lp, rp, ci, v


class PrecondOverlap(Precond):
    """isph_prec_create_overlap: ILU(k) of this rank's rows plus one layer of its neighbours' rows ("Overlap Level" 1 on
    more than one rank).  Aext: hip.Matrix of the extended subdomain (dist.extend_rows), plan: dist.HaloPlan."""

    def __init__(self, ctx, Aext, plan, level_of_fill=1, combine="add"):
        self.ctx, self.n = ctx, int(plan.nlocal)
        self.h = C.c_void_p()
        peers, sp, si, rp_ = (np.ascontiguousarray(a, dtype=np.int32) for a in (plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr))
        _check(lib().isph_prec_create_overlap(ctx.h, Aext.h, self.n, int(level_of_fill), {"add": 0, "zero": 1}[combine],
                                              len(peers), _ptr(peers), _ptr(sp), _ptr(si), _ptr(rp_), C.byref(self.h)))


def solve_block(ctx, blocks, b, x, prec=None, params=None, lda=None):
    """isph_solve_block == SolverLin_Belos::solveBlockProblem.  blocks: dim x dim nested list of Matrix / None;
    b, x column-major [lda x dim] (numpy [dim, lda] arrays or flat); x is updated in place."""
    dim = len(blocks)
    arr = (C.c_void_p * (dim * dim))()
    n = None
    for i in range(dim):
        for j in range(dim):
            Bm = blocks[i][j]
            arr[i * dim + j] = Bm.h if Bm is not None else None
            if Bm is not None and n is None:
                n = Bm.info()["nrow"]
    prm = params or SolverParams()
    info = SolveInfo()
    _check(lib().isph_solve_block(ctx.h, dim, arr, prec.h if prec is not None else None, _ptr(b), _ptr(x),
                                  n if lda is None else lda, C.byref(prm), C.byref(info), _on_device(b, x)))
    return info


class PrecondAMG(Precond):
    """isph_prec_create_amg == PrecondWrapper_ML::create() (with setNullVector when nullvec is given)."""

    def __init__(self, ctx, A, nullvec=None, params=None):
        self.ctx, self.n = ctx, A.info()["nrow"]
        self.h = C.c_void_p()
        prm = params or AmgParams()
        nv = None if nullvec is None else _f64(nullvec)
        _check(lib().isph_prec_create_amg(ctx.h, A.h, C.byref(prm), _ptr(nv), int(nv is not None and _is_torch(nv)),
                                          C.byref(self.h)))

    @property
    def levels(self):
        return lib().isph_prec_amg_levels(self.h)

    def level_info(self, l):
        a = (C.c_longlong * 3)()
        _check(lib().isph_prec_amg_info(self.ctx.h, self.h, l, a))
        return dict(rows=int(a[0]), nnz=int(a[1]), nnz_p=int(a[2]))

    def export(self, l, what="A"):
        i = self.level_info(l)
        nnz = i["nnz"] if what == "A" else i["nnz_p"]
        rp = np.zeros(i["rows"] + 1, dtype=np.int32)
        ci = np.zeros(nnz, dtype=np.int32)
        v = np.zeros(nnz)
        _check(lib().isph_prec_amg_export(self.ctx.h, self.h, l, 0 if what == "A" else 1, _ptr(rp), _ptr(ci), _ptr(v)))
        return rp, ci, v

    def aggregates(self, l):
        a = np.zeros(self.level_info(l)["rows"], dtype=np.int32)
        _check(lib().isph_prec_amg_aggregates(self.ctx.h, self.h, l, _ptr(a)))
        return a


def solve(ctx, A, b, x, prec=None, singular=False, null_mask=None, params=None, nvec=1, lda=None):
    """isph_solve == SolverLin_Belos::solveProblem.  b and x are updated in
    place (b by its projection when singular); returns SolveInfo."""
    prm = params or SolverParams()
    info = SolveInfo()
    n = A.info()["nrow"]
    mask = None if null_mask is None else np.ascontiguousarray(null_mask, dtype=np.int32)
    ld = n if lda is None else lda
    _need(b, ld * (nvec - 1) + n, "b [lda x nvec]"); _need(x, ld * (nvec - 1) + n, "x [lda x nvec]"); _need(mask, n, "null_mask [n]")
    _check(lib().isph_solve(ctx.h, A.h, prec.h if prec is not None else None, _ptr(b), _ptr(x), nvec,
                            n if lda is None else lda, int(singular), _ptr(mask), C.byref(prm), C.byref(info),
                            _on_device(b, x)))
    return info


def _need(a, count, what):
    """operand of a C-ABI call: at least `count` elements (None passes)"""
    if a is None:
        return
    have = int(a.numel()) if _is_torch(a) else int(np.asarray(a).size)
    if have < count:
        raise ValueError("%s: %d elements given, %d needed" % (what, have, count))


def particles_view(parts, colmap, kernel="wendland", kinds=None, vfrac=None, Gc=None, Lc=None, keep=None,
                   pnd=None, morris_safe_coeff=0.43301, normal=None, solid_normal_diag=1.0):
    """Builds the isph_particles struct over host (numpy) or device (torch)
    arrays.  `keep` collects references so the buffers outlive the call."""
    keep = keep if keep is not None else []
    if kinds is not None:
        ntypes = len(kinds)
    elif int(parts["nall"]) == 0:                                  # a rank without particles (LAMMPS allows empty subdomains)
        ntypes = 1
    else:
        ntypes = int(np.max(parts["type"])) if not _is_torch(parts["type"]) else int(parts["type"].max().item())
    kind = np.ascontiguousarray([0] + list(kinds if kinds is not None else [99] * ntypes), dtype=np.int32)
    h = np.full((ntypes + 1, ntypes + 1), float(parts["h"]))
    cutsq = np.full((ntypes + 1, ntypes + 1), float(parts["cut"]) ** 2)
    x, typ = _f64(parts["x"]), _i32(parts["type"])
    nidx, cm = _i32(parts["neigh_idx"]), _i32(colmap)
    nptr_raw = parts["neigh_ptr"]
    wide = str(nptr_raw.dtype) in ("int64", "torch.int64")        # 64-bit list offsets -> isph_particles::neigh_ptr64
    nptr = (nptr_raw if _is_torch(nptr_raw) else np.ascontiguousarray(nptr_raw)) if wide else _i32(nptr_raw)
    pnd = None if pnd is None else _f64(pnd)
    normal = None if normal is None else _f64(normal)
    keep += [kind, h, cutsq, x, typ, nptr, nidx, cm, vfrac, Gc, Lc, pnd, normal]
    nl, na, dm = int(parts["nlocal"]), int(parts["nall"]), int(parts["dim"])
    # the C ABI takes bare pointers: a short array would be read past its end
    _need(x, 3 * na, "x [nall][3]"); _need(typ, na, "type [nall]"); _need(cm, na, "colmap [nall]")
    _need(nptr, nl + 1, "neigh_ptr [nlocal+1]"); _need(vfrac, na, "vfrac [nall]"); _need(pnd, na, "pnd [nall]")
    _need(Gc, nl * dm * dm, "Gc [nlocal][dim*dim]"); _need(Lc, nl * dm * (dm + 1) // 2, "Lc [nlocal][dimL]")
    _need(normal, 3 * na, "normal [nall][3]")
    pv = _Particles(int(parts["dim"]), int(parts["nlocal"]), int(parts["nall"]), ntypes, KERNELS[kernel],
                    _ptr(x), _ptr(typ), _ptr(kind), _ptr(h), _ptr(cutsq), None if wide else _ptr(nptr), _ptr(nidx), _ptr(cm),
                    _ptr(vfrac), _ptr(Gc), _ptr(Lc), int(pnd is not None), _ptr(pnd), float(morris_safe_coeff),
                    _ptr(normal), float(solid_normal_diag), _ptr(nptr) if wide else None)
    return pv, _on_device(x, typ, nptr, nidx, cm, vfrac, Gc, Lc, pnd, normal), keep


def assemble_poisson(ctx, parts, colmap, dt, rho, vstar, antisym=True, singular=NULLSPACE, rank0=True,
                     ncol=None, vfrac=None, kernel="wendland", b_out=None, kinds=None, pnd=None, Gc=None, Lc=None,
                     morris_safe_coeff=0.43301, normal=None, solid_normal_diag=1.0):
    """isph_assemble_poisson == PairISPH_Corrected::computePoisson."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, keep=keep, kinds=kinds, pnd=pnd, Gc=Gc,
                                   Lc=Lc, morris_safe_coeff=morris_safe_coeff, normal=normal,
                                   solid_normal_diag=solid_normal_diag)
    rho, vstar = _f64(rho), _f64(vstar)
    nlocal = int(parts["nlocal"])
    _need(rho, int(parts["nall"]), "rho [nall]"); _need(vstar, 3 * int(parts["nall"]), "vstar [nall][3]")
    if b_out is None:
        if dev:
            import torch
            b_out = torch.zeros(nlocal, dtype=torch.float64, device=rho.device)
        else:
            b_out = np.zeros(nlocal)
    A = Matrix(ctx)
    _check(lib().isph_assemble_poisson(ctx.h, C.byref(pv), int(antisym), float(dt), _ptr(rho), _ptr(vstar),
                                       singular, int(rank0), nlocal if ncol is None else ncol, C.byref(A.h),
                                       _ptr(b_out), dev))
    return A, b_out


def assemble_helmholtz(ctx, parts, colmap, dt, theta, nu, rho, pres, force, g, vel, antisym=True, incremental=True,
                       ncol=None, vfrac=None, Gc=None, Lc=None, kernel="wendland", kinds=None, pnd=None,
                       morris_safe_coeff=0.43301, rhs_only=False):
    """isph_assemble_helmholtz == PairISPH_Corrected::computeHelmholtz.  Returns (Matrix, b) with b
    column-major [nlocal x dim] flattened (component k at b[k*nlocal:(k+1)*nlocal])."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, Lc=Lc, keep=keep, kinds=kinds,
                                   pnd=pnd, morris_safe_coeff=morris_safe_coeff)
    nu, rho, pres, force, vel = map(_f64, (nu, rho, pres, force, vel))
    gv = np.ascontiguousarray(g, dtype=np.float64)
    nlocal, dim = int(parts["nlocal"]), int(parts["dim"])
    na = int(parts["nall"])
    for a_, c_, w_ in ((nu, na, "nu [nall]"), (rho, na, "rho [nall]"), (pres, na, "pres [nall]"), (force, 3 * na, "force [nall][3]"),
                       (vel, 3 * na, "v [nall][3]")):
        _need(a_, c_, w_)
    if dev:
        import torch
        b_out = torch.zeros(nlocal * dim, dtype=torch.float64, device=rho.device)
    else:
        b_out = np.zeros(nlocal * dim)
    A = None if rhs_only else Matrix(ctx)
    _check(lib().isph_assemble_helmholtz(ctx.h, C.byref(pv), int(antisym), float(dt), float(theta), _ptr(nu), _ptr(rho),
                                         _ptr(pres), _ptr(force), _ptr(gv), int(incremental), _ptr(vel),
                                         nlocal if ncol is None else ncol, None if rhs_only else C.byref(A.h),
                                         _ptr(b_out), nlocal, dev))
    return A, b_out


def assemble_solute_transport(ctx, parts, colmap, dt, theta, dcoeff, conc, antisym=True, ncol=None, vfrac=None, Gc=None,
                              Lc=None, kernel="wendland", kinds=None):
    """isph_assemble_solute_transport == PairISPH_Corrected::computeSoluteTransportSpecies.  Returns (Matrix, b[nlocal])."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, Lc=Lc, keep=keep, kinds=kinds)
    conc = _f64(conc)
    nlocal = int(parts["nlocal"])
    _need(conc, int(parts["nall"]), "conc [nall]")
    if dev:
        import torch
        b_out = torch.zeros(nlocal, dtype=torch.float64, device=conc.device)
    else:
        b_out = np.zeros(nlocal)
    A = Matrix(ctx)
    _check(lib().isph_assemble_solute_transport(ctx.h, C.byref(pv), int(antisym), float(dt), float(theta), float(dcoeff),
                                                _ptr(conc), nlocal if ncol is None else ncol, C.byref(A.h), _ptr(b_out), dev))
    return A, b_out


def assemble_applied_potential(ctx, parts, colmap, sigma, phi, antisym=True, ncol=None, vfrac=None, Gc=None, Lc=None,
                               kernel="wendland", kinds=None):
    """isph_assemble_applied_potential == PairISPH_Corrected::computeAppliedElectricPotential.  Returns (Matrix, b[nlocal])."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, Lc=Lc, keep=keep, kinds=kinds)
    phi = _f64(phi)
    sg = None if sigma is None else _f64(sigma)
    nlocal = int(parts["nlocal"])
    _need(phi, int(parts["nall"]), "phi [nall]"); _need(sg, int(parts["nall"]), "sigma [nall]")
    if dev:
        import torch
        b_out = torch.zeros(nlocal, dtype=torch.float64, device=phi.device)
    else:
        b_out = np.zeros(nlocal)
    A = Matrix(ctx)
    _check(lib().isph_assemble_applied_potential(ctx.h, C.byref(pv), int(antisym), _ptr(sg), _ptr(phi),
                                                 nlocal if ncol is None else ncol, C.byref(A.h), _ptr(b_out), dev))
    return A, b_out


def assemble_block_helmholtz(ctx, parts, colmap, dt, theta, beta, nu, rho, pres, force, g, vel, normal=None, antisym=True,
                             incremental=True, ncol=None, vfrac=None, Gc=None, Lc=None, kernel="wendland", kinds=None,
                             pnd=None, morris_safe_coeff=0.43301):
    """isph_assemble_block_helmholtz == PairISPH_Corrected::computeBlockHelmholtz.  Returns (blocks, b): blocks is a
    dim x dim nested list of Matrix / None, b column-major [nlocal x dim] flattened."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, Lc=Lc, keep=keep, kinds=kinds,
                                   pnd=pnd, morris_safe_coeff=morris_safe_coeff)
    nu, rho, pres, force, vel = map(_f64, (nu, rho, pres, force, vel))
    nrm = None if normal is None else _f64(normal)
    na = int(parts["nall"])
    for a_, c_, w_ in ((nu, na, "nu [nall]"), (rho, na, "rho [nall]"), (pres, na, "pres [nall]"), (force, 3 * na, "force [nall][3]"),
                       (vel, 3 * na, "v [nall][3]"), (nrm, 3 * na, "normal [nall][3]")):
        _need(a_, c_, w_)
    gv = np.ascontiguousarray(g, dtype=np.float64)
    nlocal, dim = int(parts["nlocal"]), int(parts["dim"])
    if dev:
        import torch
        b_out = torch.zeros(nlocal * dim, dtype=torch.float64, device=rho.device)
    else:
        b_out = np.zeros(nlocal * dim)
    hs = (C.c_void_p * (dim * dim))()
    _check(lib().isph_assemble_block_helmholtz(ctx.h, C.byref(pv), int(antisym), float(dt), float(theta), float(beta),
                                               _ptr(nu), _ptr(rho), _ptr(pres), _ptr(force), _ptr(gv), int(incremental),
                                               _ptr(vel), _ptr(nrm), nlocal if ncol is None else ncol, hs, _ptr(b_out),
                                               nlocal, dev))
    blocks = [[None] * dim for _ in range(dim)]
    for i in range(dim):
        for j in range(dim):
            if hs[i * dim + j]:
                Bm = Matrix(ctx)
                Bm.h = C.c_void_p(hs[i * dim + j])
                blocks[i][j] = Bm
    return blocks, b_out


def compute_volumes(ctx, parts, colmap, kernel="wendland"):
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, keep=keep)
    nlocal = int(parts["nlocal"])
    if dev:
        import torch
        out = torch.zeros(nlocal, dtype=torch.float64, device=parts["x"].device)
    else:
        out = np.zeros(nlocal)
    _check(lib().isph_compute_volumes(ctx.h, C.byref(pv), _ptr(out), dev))
    return out


def compute_pnd(ctx, parts, colmap, kernel="wendland", kinds=None):
    """isph_compute_pnd: particle number density of the MorrisHolmes mirror (functor_normal.h:57-133); [nlocal]."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, keep=keep, kinds=kinds)
    nlocal = int(parts["nlocal"])
    if dev:
        import torch
        out = torch.zeros(nlocal, dtype=torch.float64, device=parts["x"].device)
    else:
        out = np.zeros(nlocal)
    _check(lib().isph_compute_pnd(ctx.h, C.byref(pv), _ptr(out), dev))
    return out


def compute_corrections(ctx, parts, colmap, vfrac, kernel="wendland"):
    """isph_compute_corrections == computeGradientCorrection + computeLaplacianCorrection.
    Returns (Gc [nlocal, dim*dim], Lc [nlocal, dimL])."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, keep=keep)
    nlocal, dim = int(parts["nlocal"]), int(parts["dim"])
    dL = dim * (dim + 1) // 2
    if dev:
        import torch
        G = torch.zeros((nlocal, dim * dim), dtype=torch.float64, device=parts["x"].device)
        Lc = torch.zeros((nlocal, dL), dtype=torch.float64, device=parts["x"].device)
    else:
        G, Lc = np.zeros((nlocal, dim * dim)), np.zeros((nlocal, dL))
    _check(lib().isph_compute_corrections(ctx.h, C.byref(pv), _ptr(G), _ptr(Lc), dev))
    return G, Lc


def _out_like(dev, ref, shape):
    """output buffer on the side the inputs live on (device tensors in -> device tensor out)"""
    if dev:
        import torch
        return torch.zeros(shape, dtype=torch.float64, device=ref.device)
    return np.zeros(shape)


def _same_side(dev, *arrays):
    """every operand of a call must live on the side the particle arrays live on"""
    for a in arrays:
        if a is not None and bool(_is_torch(a)) != bool(dev):
            raise ValueError("mixing host and device operands in one call")


def gradient(ctx, parts, colmap, f, vfrac, antisym=True, alpha=1.0, filt=None, Gc=None, kernel="wendland", kinds=None):
    """isph_gradient: scalar field f [nall] -> [nlocal, 3]."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, keep=keep, kinds=kinds)
    f = _f64(f)
    _same_side(dev, f)
    _need(f, int(parts["nall"]), "f [nall]")
    out = _out_like(dev, f, (int(parts["nlocal"]), 3))
    fi, fj = filt if filt is not None else (127, 127)
    _check(lib().isph_gradient(ctx.h, C.byref(pv), int(antisym), _ptr(f), float(alpha), int(filt is not None), fi, fj,
                               _ptr(out), dev))
    return out


def divergence(ctx, parts, colmap, f, vfrac, antisym=True, alpha=1.0, filt=None, Gc=None, kernel="wendland", kinds=None):
    """isph_divergence: vector field f [nall, 3] -> [nlocal]."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, keep=keep, kinds=kinds)
    f = _f64(f)
    _same_side(dev, f)
    _need(f, 3 * int(parts["nall"]), "f [nall][3]")
    out = _out_like(dev, f, (int(parts["nlocal"]),))
    fi, fj = filt if filt is not None else (127, 127)
    _check(lib().isph_divergence(ctx.h, C.byref(pv), int(antisym), _ptr(f), float(alpha), int(filt is not None), fi, fj,
                                 _ptr(out), dev))
    return out


def correct_velocity_pressure(ctx, parts, colmap, dt, rho, dp, vstar, p, vfrac, antisym=True, incremental=True, Gc=None,
                              kernel="wendland"):
    """in-place on vstar [nall,3] and p [nall] (numpy arrays or device tensors, like the particle arrays)."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, keep=keep)
    rho, dp = _f64(rho), _f64(dp)
    _same_side(dev, rho, dp, vstar, p)
    na = int(parts["nall"])
    _need(rho, na, "rho [nall]"); _need(dp, na, "dp [nall]"); _need(vstar, 3 * na, "vstar [nall][3]"); _need(p, na, "p [nall]")
    _check(lib().isph_correct_velocity_pressure(ctx.h, C.byref(pv), int(antisym), float(dt), _ptr(rho), _ptr(dp),
                                                _ptr(vstar), _ptr(p), int(incremental), dev))


def advance_begin(ctx, parts, colmap, dt, p, v, vnp1, vfrac, antisym=True, Gc=None, kernel="wendland"):
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, keep=keep)
    p, v, vnp1 = _f64(p), _f64(v), _f64(vnp1)
    _same_side(dev, p, v, vnp1)
    na = int(parts["nall"])
    _need(p, na, "p [nall]"); _need(v, 3 * na, "v [nall][3]"); _need(vnp1, 3 * na, "vnp1 [nall][3]")
    out = _out_like(dev, p, (int(parts["nlocal"]),))
    _check(lib().isph_advance_begin(ctx.h, C.byref(pv), int(antisym), float(dt), _ptr(p), _ptr(v), _ptr(vnp1), _ptr(out),
                                    dev))
    return out


def advance_end(ctx, count, dim, dt, dp, vnp1, p, x, v):
    """in-place on p [count], x [count,3], v [count,3] (all numpy or all device tensors)."""
    dp, vnp1 = _f64(dp), _f64(vnp1)
    dev = int(bool(_is_torch(p)))
    _same_side(dev, dp, vnp1, p, x, v)
    _check(lib().isph_advance_end(ctx.h, int(count), int(dim), float(dt), _ptr(dp), _ptr(vnp1), _ptr(p), _ptr(x),
                                  _ptr(v), dev))


def compute_shift(ctx, parts, colmap, alpha, shiftcut, nonfluidweight, kernel="wendland", kinds=None):
    """isph_compute_shift -> dr [nlocal, 3]."""
    keep = []
    if _is_torch(parts["x"]):
        import torch
        ones = torch.ones(int(parts["nall"]), dtype=torch.float64, device=parts["x"].device)
    else:
        ones = np.ones(int(parts["nall"]))
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=ones, keep=keep, kinds=kinds)
    dr = _out_like(dev, parts["x"], (int(parts["nlocal"]), 3))
    _check(lib().isph_compute_shift(ctx.h, C.byref(pv), float(alpha), float(shiftcut), float(nonfluidweight), _ptr(dr),
                                    dev))
    return dr


def apply_shift(ctx, parts, colmap, dr, x, v, p, vfrac, antisym=True, fixed=None, Gc=None, kernel="wendland", kinds=None):
    """isph_apply_shift: in place on x, v [nall,3] and p [nall] (numpy)."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, keep=keep, kinds=kinds)
    fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.int32)
    dr = _f64(dr)
    _same_side(dev, dr, x, v, p)
    na = int(parts["nall"])
    _need(dr, 3 * int(parts["nlocal"]), "dr [nlocal][3]"); _need(x, 3 * na, "x [nall][3]"); _need(v, 3 * na, "v [nall][3]"); _need(p, na, "p [nall]")
    _check(lib().isph_apply_shift(ctx.h, C.byref(pv), int(antisym), _ptr(fx), _ptr(dr), _ptr(x), _ptr(v), _ptr(p), dev))


def shift_particles(ctx, parts, colmap, shift, shiftcut, nonfluidweight, dt, x, v, p, vfrac, antisym=True, fixed=None,
                    Gc=None, kernel="wendland", kinds=None):
    """isph_shift_particles: in place on x, v, p; returns the max fluid speed used."""
    keep = []
    pv, dev, keep = particles_view(parts, colmap, kernel=kernel, vfrac=vfrac, Gc=Gc, keep=keep, kinds=kinds)
    fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.int32)
    vmax = C.c_double(0.0)
    _same_side(dev, x, v, p)
    na = int(parts["nall"])
    _need(x, 3 * na, "x [nall][3]"); _need(v, 3 * na, "v [nall][3]"); _need(p, na, "p [nall]")
    _check(lib().isph_shift_particles(ctx.h, C.byref(pv), int(antisym), _ptr(fx), float(shift), float(shiftcut),
                                      float(nonfluidweight), float(dt), _ptr(x), _ptr(v), _ptr(p), C.byref(vmax), dev))
    return vmax.value
