"""Run by tests/test_gpu_parity.py::test_entry_points_read_no_more_than_the_abi_documents in a child process.

Every host operand of the main entry points is placed so that it ENDS at an unreadable page (anonymous mmap + mprotect):
an entry point that reads one element more than include/isph_hip.h documents dies with SIGSEGV here instead of reading
whatever follows the caller's buffer (the [nall]-for-[nlocal] staging of Gc / Lc did exactly that, intermittently).
Exit code 0 = every call came back.  `--negative-control`: a call that IS one element short, to show the guard bites.
"""
import ctypes
import mmap
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import isph_amd  # noqa: F401,E402
from isph_amd import hip, workload  # noqa: E402

_libc = ctypes.CDLL(None, use_errno=True)
_keep = []


def guarded(a):
    a = np.ascontiguousarray(a)
    if a.size == 0:
        return a
    page = mmap.PAGESIZE
    npages = (a.nbytes + page - 1) // page
    m = mmap.mmap(-1, (npages + 1) * page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    assert _libc.mprotect(ctypes.c_void_p(base + npages * page), ctypes.c_size_t(page), 0) == 0     # PROT_NONE
    g = np.frombuffer(m, dtype=a.dtype, count=a.size, offset=npages * page - a.nbytes).reshape(a.shape)
    g[...] = a
    _keep.append(m)
    return g


def main():
    ctx = hip.Context(0)
    spec = workload.TGVSpec(dim=3, ncell=(10, 10, 10), brick=(4, 4, 4), mode=workload.JITTER)
    p = dict(workload.make_tgv(spec))
    n, nall = p["nlocal"], p["nall"]
    own = p["owner_index"].astype(np.int64)
    typ = np.ones(n, np.int32)
    typ[(p["x"][:n, 1] % (2 * np.pi)) < 0.9] = 2                    # a solid slab: walls, normals, pnd
    p["type"] = typ[own]
    kinds = [99, 12]
    for k in ("x", "type", "neigh_ptr", "neigh_idx"):
        p[k] = guarded(p[k])
    colmap = guarded(workload.single_rank_colmap(p))
    G = guarded

    vf = hip.compute_volumes(ctx, p, colmap)
    vfrac = G(vf[own])
    pnd = G(hip.compute_pnd(ctx, p, colmap, kinds=kinds)[own])
    Gc, Lc = hip.compute_corrections(ctx, p, colmap, vfrac)
    Gc, Lc = G(Gc), G(Lc)                                              # [nlocal] rows, as the ABI documents
    rho, nu = G(np.ones(nall)), G(np.full(nall, 0.1))
    vel = G(np.ascontiguousarray(p["v"]))
    pres, force, g3 = G(np.cos(p["x"][:, 0])), G(np.zeros((nall, 3))), np.zeros(3)
    nrm = np.zeros((nall, 3))
    nrm[np.asarray(p["type"]) == 2, 1] = 1.0
    nrm = G(nrm)

    for antisym in (True, False):
        kw = dict(vfrac=vfrac, kinds=kinds, Gc=None if antisym else Gc, Lc=None if antisym else Lc)
        A, b = hip.assemble_poisson(ctx, p, colmap, spec.dt, rho, vel, antisym=antisym, **kw)
        hip.assemble_poisson(ctx, p, colmap, spec.dt, rho, vel, antisym=antisym, vfrac=vfrac, kinds=kinds, Gc=Gc, Lc=Lc, pnd=pnd,
                             normal=nrm)
        H, bh = hip.assemble_helmholtz(ctx, p, colmap, spec.dt, 0.5, nu, rho, pres, force, g3, vel, antisym=antisym, pnd=pnd, **kw)
        hip.assemble_block_helmholtz(ctx, p, colmap, spec.dt, 0.5, 0.1, nu, rho, pres, force, g3, vel, normal=nrm, antisym=antisym,
                                     vfrac=vfrac, kinds=kinds, Gc=Gc, Lc=Lc)
        hip.assemble_solute_transport(ctx, p, colmap, spec.dt, 0.5, 0.3, pres, antisym=antisym, **kw)
        hip.assemble_applied_potential(ctx, p, colmap, nu, pres, antisym=antisym, **kw)
        hip.gradient(ctx, p, colmap, pres, vfrac, antisym=antisym, Gc=kw["Gc"], kinds=kinds)
        hip.divergence(ctx, p, colmap, vel, vfrac, antisym=antisym, Gc=kw["Gc"], kinds=kinds)
        # the solves: three right-hand sides in one [lda x 3] view, SpMV, preconditioner application
        M = hip.Precond(ctx, H, "bjacobi-ilu0", 256)
        xg, bg = G(np.zeros(3 * n)), G(np.asarray(bh))
        info = hip.solve(ctx, H, bg, xg, prec=M, nvec=3, lda=n)
        assert info.converged == 1
        # a padded view: lda > n, the arrays own exactly lda (nvec - 1) + n elements
        lda = n + 5
        bpad, xpad = np.zeros(2 * lda + n), np.zeros(2 * lda + n)
        for k in range(3):
            bpad[k * lda:k * lda + n] = np.asarray(bh)[k * n:(k + 1) * n]
        bpad, xpad = G(bpad), G(xpad)
        assert hip.solve(ctx, H, bpad, xpad, prec=M, nvec=3, lda=lda).converged == 1
        for k in range(3):
            assert np.max(np.abs(xpad[k * lda:k * lda + n] - xg[k * n:(k + 1) * n])) <= 1e-12 * max(np.abs(xg).max(), 1e-300)
        H.spmv(G(np.ones(n)))
        M.apply(G(np.ones(n)))
        Ms = hip.PrecondAMG(ctx, A, nullvec=G(np.full(n, 1.0 / np.sqrt(n))))
        x1, b1 = G(np.zeros(n)), G(np.asarray(b))
        assert hip.solve(ctx, A, b1, x1, prec=Ms, singular=True, null_mask=G(np.ones(n, np.int32))).converged == 1
        rp, ci, val = A.export_csr()
        hip.Matrix.from_csr(ctx, G(rp), G(ci), G(val)).spmv(G(np.ones(n)))
        # the fused host ingress (matrix + block ILU(0) in one call) and the ranged row export
        Af, Mf = hip.Matrix.from_host_csr_with_bjacobi(ctx, G(rp), G(ci), G(val), 256)
        Mf.apply(G(np.ones(n)))
        rpr, cir, vr = Af.export_rows(n - 70, 70)
        assert rpr[-1] == rp[n] - rp[n - 70] and np.array_equal(cir, ci[rp[n - 70]:rp[n]])
        # the streaming operators either side of the solve (in place on their [nall] operands)
        vs, pp = G(np.ascontiguousarray(p["v"])), G(np.asarray(pres).copy())
        hip.correct_velocity_pressure(ctx, p, colmap, 0.01, rho, pres, vs, pp, vfrac, antisym=antisym, Gc=kw["Gc"])
        dpa = hip.advance_begin(ctx, p, colmap, 0.01, pres, vel, vs, vfrac, antisym=antisym, Gc=kw["Gc"])
        xg2, vg2, pg2 = G(np.ascontiguousarray(p["x"][:n])), G(np.ascontiguousarray(p["v"][:n])), G(np.asarray(pres)[:n].copy())
        hip.advance_end(ctx, n, 3, 0.01, G(dpa), G(np.ascontiguousarray(vs[:n])), pg2, xg2, vg2)
        xs, vsh, psh = G(np.array(p["x"])), G(np.ascontiguousarray(p["v"])), G(np.asarray(pres).copy())
        hip.shift_particles(ctx, p, colmap, 0.05, 0.8 * p["cut"], 0.7, spec.dt, xs, vsh, psh, vfrac, antisym=antisym, Gc=kw["Gc"],
                            kinds=kinds, fixed=[0, 0, 1])
        # the reference's own decomposition (one subdomain, level-scheduled) and the recycling solver
        Mw = hip.PrecondSchwarz(ctx, H, level_of_fill=0, overlap=0, block_size=0)
        Mw.apply(G(np.ones(n)))
        xr, br = G(np.zeros(n)), G(np.asarray(bh)[:n].copy())
        assert hip.solve(ctx, H, br, xr, prec=M, params=hip.SolverParams(solver_type=2, num_blocks=20, num_recycled=5)).converged == 1
    print("guarded operands: every entry point stayed inside its buffers")
    ctx.close()


def negative_control():
    """one element short on purpose, past the element-count check of the ctypes layer: must die with SIGSEGV"""
    import ctypes as C
    import resource
    resource.setrlimit(resource.RLIMIT_CORE, (0, 0))      # the crash is the point; no core file
    ctx = hip.Context(0)
    n = 5000
    A = hip.Matrix.from_csr(ctx, np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.ones(n))
    x, y = guarded(np.ones(n - 1)), np.zeros(n)
    rc = hip.lib().isph_spmv(ctx.h, A.h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 0)
    print("over-read went unnoticed, rc", rc)


if __name__ == "__main__":
    negative_control() if "--negative-control" in sys.argv else main()
