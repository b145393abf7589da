#!/usr/bin/env python3
"""bench.py -- pressure-Poisson solves/sec + SpMV roofline on the 3-D Taylor-Green
vortex, one process per GPU.

A "step" = one pass of the reference's `ISPH: solvePoisson` scope
(pair_isph.cpp:1008-1012 -> solver_lin_belos.h:130-222) over the resident
system: preconditioner build (rebuilt every solve like prec->create()/free())
+ right-preconditioned FGMRES(50)/DGKS + null-space projections.  The matrix
and right-hand side are assembled on the GPU before the timed region
(`ISPH: computePoisson`, reported separately as assemble_ms).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def spmv_algorithmic_bytes(nrow, nnz):
    """12 B per stored entry (fp64 value + int32 column),
    x read once, y written once, row pointers."""
    return 12 * nnz + 16 * nrow + 4 * (nrow + 1)


def solve_roofline(mat, pinfo, inf, prm, nprof, prof, sec_per_solve, prec):
    """SURVEY.md 8(d) summed over the whole solve, and the kernels of the hot loop one by one.

    Algorithmic bytes (N rows, nnz matrix entries, nnz_f factor entries; FGMRES(m) with the null vector deflated inside
    the Gram-Schmidt step, so a step at column j of a cycle projects against nk = j + 2 vectors):
      SpMV                     12 nnz + 16 N + 4 (N + 1)
      preconditioner apply     12 nnz_f + 16 N                    (block ILU(0): two triangular sweeps over the factor)
      Gram-Schmidt step        dots (nk + 1) 8N, update (nk + 1) 8N + 8N, norm 8N, flexible store of z_j 8N; dots and
                               update once more when the DGKS test asks for the second pass (counted from the run)
      per cycle                r = b - A x: SpMV + 3 vectors; x += Z y: (j + 2) 8N
      set-up                   read A (12 nnz + 4 (N + 1)), write the factor (12 nnz_f + 8 N)
    The kernels: algorithmic bytes of what one launch has to touch / its average duration from HIP events on the
    library's stream (profile passes outside the timed region; isph_ctx_profile_read)."""
    N, nnz = mat["nrow"], mat["nnz"]
    nnzf = int(pinfo.get("factor_nnz", 0)) if prec.startswith("bjacobi-ilu") else 0
    v8 = 8.0 * N
    spmv_b = 12.0 * nnz + 16.0 * N + 4.0 * (N + 1)
    prec_b = (12.0 * nnzf + 16.0 * N) if nnzf else (24.0 * N if prec == "jacobi" else 0.0)
    m = prm.num_blocks
    iters, cycles = int(inf.iters), int(inf.restarts) + 1
    gs = dot_b = axd_b = axn_b = 0.0
    per_cycle = 0.0
    left = iters
    for _ in range(cycles):
        jn = min(m, left)
        left -= jn
        for j in range(jn):
            nk = j + 2
            gs += (nk + 1) * v8 + (nk + 1) * v8 + v8 + v8 + v8
            dot_b += (nk + 1) * v8                     # k_multi_dot: V[0..nk) and w
            axd_b += (nk + 2) * v8                     # k_multi_axpy_dot: V, w in, w out (the second projection rides along)
            axn_b += 2 * v8                            # k_multi_axpy_norm without the second pass: w in, v_next out
        per_cycle += spmv_b + 3 * v8 + (jn + 2) * v8
    reorth = int(getattr(inf, "reorth", 0))
    avg_nk = (iters / cycles) / 2.0 + 2.0
    gs += reorth * 2 * (avg_nk + 1) * v8
    axn_b += reorth * (avg_nk + 1) * v8
    setup_b = (12.0 * nnz + 4.0 * (N + 1)) + (12.0 * nnzf + 8.0 * N) if nnzf else 0.0
    total = iters * (spmv_b + prec_b) + gs + per_cycle + setup_b
    kern = {}

    def add(name, cls, bytes_per_solve):
        ms, calls = prof.get(cls, (0.0, 0))
        if calls:
            per_launch = bytes_per_solve * nprof / calls
            kern[name] = {"avg_us": ms / calls * 1e3, "launches_per_solve": calls / nprof, "algorithmic_bytes": per_launch,
                          "GBps": per_launch / (ms / calls * 1e-3) / 1e9, "frac": per_launch / (ms / calls * 1e-3) / 1e9 / HBM_PEAK_GBS}

    add("k_sell_spmv16", "spmv", (iters + cycles + 1) * spmv_b)           # + r = b - A x per cycle + the explicit residual
    if nnzf:
        add("k_ilu_solve_stream", "prec_apply", iters * prec_b)
        add("k_ilu_extract", "ilu_extract", 12.0 * nnz + 12.0 * nnzf)
        add("k_ilu_schedule", "ilu_schedule", 4.0 * nnzf + 10.0 * nnzf)     # pattern in, stream words + destinations out
        add("k_ilu_factor", "ilu_factor", 12.0 * nnzf + 10.0 * nnzf + 8.0 * N)
    add("k_multi_dot", "multi_dot", dot_b)
    add("k_multi_axpy_dot", "multi_axpy_dot", axd_b)
    add("k_multi_axpy_norm", "multi_axpy_norm", axn_b)
    in_kernels_ms = sum(ms for ms, _ in prof.values()) / nprof
    return {"solve": {"algorithmic_bytes": total, "ms": sec_per_solve * 1e3, "GBps": total / sec_per_solve / 1e9,
                      "frac": total / sec_per_solve / 1e9 / HBM_PEAK_GBS, "iterations": iters, "cycles": cycles,
                      "second_gram_schmidt_passes": reorth,
                      "bytes": {"spmv": iters * spmv_b, "prec_apply": iters * prec_b, "gram_schmidt": gs, "per_cycle": per_cycle,
                                "setup": setup_b},
                      "ms_inside_bracketed_kernels": in_kernels_ms},
            "kernels": kern}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncell", type=int, default=100, help="lattice cells per axis PER GPU brick edge")
    ap.add_argument("--mode", default="advect", choices=["advect", "jitter", "lattice"])
    ap.add_argument("--prec", default="bjacobi-ilu0", choices=["none", "jacobi", "bjacobi-ilu0", "bjacobi-ilu1", "bjacobi-ilu2", "sa-amg", "ilu0", "ilu1", "overlap-ilu0", "overlap-ilu1",
                             "schwarz-ilu0", "schwarz-ilu1"],
                    help="bjacobi-ilu<k>: block stream (production); ilu<k>: ILU(k) of the whole local matrix = Ifpack on one "
                         "rank; schwarz-ilu<k>: --block rows per subdomain + --overlap layers, level-scheduled (fidelity path)")
    ap.add_argument("--overlap", type=int, default=1, help='schwarz-ilu<k>: "Overlap Level" (precond_ifpack.h:43)')
    ap.add_argument("--combine", default="add", choices=["add", "zero"], help='schwarz-ilu<k>: "schwarz: combine mode"')
    ap.add_argument("--amg-theta", type=float, default=0.0, help='"aggregation: threshold" of the sa-amg variant (ML default 0)')
    ap.add_argument("--amg-smoother", default="ml.xml", choices=["ml.xml", "symmetric"],
                    help='step workload with --prec sa-amg: "ml.xml" = Gauss-Seidel, efficient symmetric (forward sweeps before, backward sweeps '
                         'after the coarse correction: bench-script/hopper/tgv/1728/ml.xml); "symmetric" = symmetric sweeps on both sides')
    ap.add_argument("--block", type=int, default=512)
    ap.add_argument("--brick", default="10,10,5",
                    help="particle numbering: bricks of this many lattice cells, brick by brick (x fastest inside).  When the "
                         "bricks tile the lattice and hold <= 1024 particles they are the block-Jacobi subdomains "
                         "(isph_prec_create_blocks), like the bricks of LAMMPS' decomposition are Ifpack's; otherwise "
                         "subdomains are --block consecutive rows")
    ap.add_argument("--order", default="lexicographic", choices=["bricks", "lexicographic", "sortbin", "shuffled"],
                    help="the atom order the particles are handed over in (pair_isph.cpp:1258-1259: the reference's rows "
                         "follow LAMMPS' atom order): lexicographic = create_atoms on the lattice, x fastest (default); sortbin "
                         "= LAMMPS' atom sorting, bins of half the neighbour cutoff (atom_modify sort); shuffled = a random "
                         "permutation (after migration); bricks = --brick cells, brick by brick (the numbering of round 4)")
    ap.add_argument("--library-order", default="on", choices=["on", "off"],
                    help="on (default): the library numbers the matrix rows itself (isph_ctx_set_ordering BRICKS) and its "
                         "bricks are the block-Jacobi subdomains; off: rows in the caller's atom order, subdomains = the "
                         "generator's bricks when --order bricks tiles the lattice, else --block consecutive rows")
    ap.add_argument("--no-orders", action="store_true", help="skip the table of the other atom orders beside the headline")
    ap.add_argument("--no-step", action="store_true", help="skip the 3-step time-step leg beside the headline")
    ap.add_argument("--colour", type=int, default=0,
                    help="numbering inside a brick: 0 lexicographic, c > 1 multi-colour with period c (isph_workload.h)")
    ap.add_argument("--kernel", default="wendland", choices=["wendland", "quintic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the SolverLin drop-in leg (host CSR through the C++ mirror)")
    ap.add_argument("--no-alt", action="store_true", help="skip the jacobi / sa-amg lines measured beside the headline")
    ap.add_argument("--cpu-ifpack-1rank", action="store_true",
                    help="also time the reference's 1-rank configuration (whole-matrix ILU(1), one thread): minutes")
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--workload", default="solve", choices=["solve", "step"],
                    help="solve (default): the headline, one pressure-Poisson solve per step on a resident system.  step: the "
                         "reference's own benchmark protocol (bench-script/hopper/tgv/1728: tgv.xml + tgv-3d-p24.lmp) -- "
                         "consecutive ISPH time steps, everything rebuilt every step; see step_workload()")
    ap.add_argument("--theta", type=float, default=0.5, help="step workload: time discretisation of the Helmholtz step (tgv.xml:13)")
    ap.add_argument("--singular", default="nullspace", choices=["nullspace", "pinzero", "doublediag"],
                    help="step workload: \"Singular Poisson\" (sph-script/taylor-green-vortex.xml NullSpace; tgv.xml:11 of the "
                         "hopper bench asks for DoubleDiag, which it pairs with ML)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the --gpus ranks as child processes from this one (automatic when --gpus > 1 and no "
                         "torch.distributed.run environment is present)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of --gpus N on ONE device: the N process ranks all use device 0 and talk through the "
                         "host-staged transport over gloo (RCCL forms no communicator between ranks of one device); "
                         "not a performance figure")
    ap.add_argument("--force-rccl", action="store_true",
                    help="1 GPU only: route the periodic images through the RCCL halo path (send/recv to self)")
    return ap.parse_args()


METRIC_NAME = {
    "bjacobi-ilu0": "pressure-Poisson solves/sec (3D TGV, 1M particles per GPU, GMRES(50)+block-Jacobi ILU(0), tol 1e-8)",
    "bjacobi-ilu1": "pressure-Poisson solves/sec (3D TGV, 1M particles per GPU, GMRES(50)+block-Jacobi ILU(1), tol 1e-8)",
    "bjacobi-ilu2": "pressure-Poisson solves/sec (3D TGV, 1M particles per GPU, GMRES(50)+block-Jacobi ILU(2), tol 1e-8)",
    "sa-amg": "pressure-Poisson solves/sec (3D TGV, 1M particles per GPU, GMRES(50)+SA-AMG V cycle, tol 1e-8)",
}


def make_particles(args, world, rank, order):
    """One rank's brick of the 3-D TGV lattice, handed over in the atom order `order` (see --order)."""
    from isph_amd import workload
    pg = pgrid_for(world)
    n = args.ncell
    mode = {"advect": workload.ADVECT, "jitter": workload.JITTER, "lattice": workload.LATTICE}[args.mode]
    brick = tuple(int(t) for t in args.brick.split(",")) if order == "bricks" else (n, n, n)   # (n, n, n): x fastest over the rank's lattice
    spec = workload.TGVSpec(dim=3, ncell=(n * pg[0], n * pg[1], n * pg[2]), pgrid=pg, rank=rank, brick=brick,
                            colour_period=args.colour if order == "bricks" else 0, mode=mode, kernel=args.kernel,
                            cut_over_h=2.0 if args.kernel == "wendland" else 3.0)
    parts = workload.make_tgv(spec)
    if order in ("sortbin", "shuffled"):
        assert world == 1, "--order %s renumbers a single rank's particles" % order
        nl = parts["nlocal"]
        if order == "shuffled":
            q = np.random.default_rng(20251005).permutation(nl)
        else:
            # LAMMPS' atom sorting (Atom::sort / setup_sort_bins): bins of half the neighbour cutoff over the sub-domain box,
            # x fastest; atoms bin by bin, inside a bin in their previous order
            L = 2.0 * np.pi
            nb = max(1, int(L / (0.5 * spec.cut)))
            ib = np.minimum((np.mod(parts["x"][:nl], L) * (nb / L)).astype(np.int64), nb - 1)
            q = np.argsort((ib[:, 2] * nb + ib[:, 1]) * nb + ib[:, 0], kind="stable")
        parts = workload.renumber(parts, q)
    return spec, parts, brick


def pgrid_for(n):
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(n) or (n, 1, 1)


def cpu_baseline(rp, ci, val, b, block, prec, amg_theta=0.0, ifpack_1rank=False, bptr=None, keep_x=None):
    """Oracle (CPU restatement of Belos FGMRES + Ifpack ILU(k) / ML SA-AMG) timed on the host cores, every variant run
    to convergence (preconditioner set-up + the whole solve, nothing extrapolated):
      same_blocks     the GPU's own subdomains (`block` rows each), all threads
      block_per_core  SURVEY 8(d)(ii): one ILU subdomain per core = what `mpirun -np <cores>` of the reference does with
                      Ifpack AdditiveSchwarz overlap 0; level of fill 0 and the reference's default 1 (precond_ifpack.h:35)
      single_thread   same_blocks on ONE thread (the reference itself has no threading)
      ifpack_1rank    the reference on one MPI rank: ILU(1) of the whole matrix, one thread (precond_ifpack.h:35,43;
                      overlap is a no-op on one rank).  Minutes of CPU time -> only with --cpu-ifpack-1rank (the record
                      contains nothing this run did not measure).
    `value` = the fastest all-thread variant.  keep_x (a list): receives the solution of the same_blocks variant, so
    that the record can state ||x_gpu - x_cpu|| / ||x_cpu|| on the benchmark's own system."""
    import oracle as orc
    n = len(rp) - 1
    threads = orc.num_threads()

    def run(lof, bp, label):
        t0 = time.perf_counter()
        ilu = amg = None
        if prec == "sa-amg":
            amg = orc.AMG(rp, ci, val, nullvec=np.full(n, 1.0 / np.sqrt(n)), block=block, theta=amg_theta)
        elif prec.startswith("bjacobi-ilu"):
            ilu = orc.ILU(rp, ci, val, lof, bp)
        t_setup = time.perf_counter() - t0
        pk = "ilu" if ilu is not None else {"none": "none", "jacobi": "jacobi", "sa-amg": "amg"}[prec]
        t0 = time.perf_counter()
        xs, info, _ = orc.solve(rp, ci, val, b, singular=True, prec=pk, ilu=ilu, amg=amg)
        t_solve = time.perf_counter() - t0
        if keep_x is not None and not keep_x:
            keep_x.append(xs)                      # the first variant (same_blocks, all threads): the parity record
        return dict(seconds_per_solve=t_setup + t_solve, setup_s=t_setup, solve_s=t_solve, iterations=int(info.iters),
                    converged=int(info.converged), cores=orc.num_threads(), config=label, measured=True)

    lof = int(prec[-1]) if prec.startswith("bjacobi-ilu") else 0
    same = bptr if bptr is not None else np.arange(0, n + block, block).clip(0, n).astype(np.int32)
    if bptr is not None:
        block = int(np.diff(bptr).max())
    percore = np.linspace(0, n, threads + 1).astype(np.int32)
    variants = {"same_blocks": run(lof, same, "%s, %d-row blocks (the GPU's subdomains)" % (prec, block))}
    if prec.startswith("bjacobi-ilu"):
        variants["block_per_core"] = run(0, percore, "ILU(0), one subdomain per core (%d), overlap 0" % threads)
        variants["block_per_core_ilu1"] = run(1, percore, "ILU(1) = reference default fill, one subdomain per core (%d), overlap 0" % threads)
    if threads > 1:
        orc.set_num_threads(1)
        try:
            variants["single_thread"] = run(lof, same, "same_blocks on one thread")
            if ifpack_1rank and prec.startswith("bjacobi-ilu"):
                variants["ifpack_1rank"] = run(1, None, "ILU(1) of the whole matrix, one thread (reference on 1 MPI rank)")
        finally:
            orc.set_num_threads(threads)
    multi = {k: v for k, v in variants.items() if v["cores"] == threads and v["converged"]}
    best = min(multi, key=lambda k: multi[k]["seconds_per_solve"])
    out = dict(value=1.0 / multi[best]["seconds_per_solve"], unit="solves/s", cores=threads, kind="port",
               sample="the whole solve (set-up + FGMRES to 1e-8) of the same %d-row system, run to convergence; fastest "
                      "all-thread variant = %s" % (n, best),
               seconds_per_solve=multi[best]["seconds_per_solve"], fastest=best)
    out.update(variants)
    return out


def dropin_leg(A, b, repeat=5, sub_rows=0, coords=None):
    """The path `north_star` names, unchanged: the matrix as a HOST Epetra CSR handed to SolverLin_Belos::solveProblem
    (pair_isph.cpp:924-926,988-1011 -> host/solver_lin_hip.h) with PrecondWrapper_Ifpack (fill 0, overlap 0, 512-row
    subdomains = the headline preconditioner).  The C++ driver of the mirror classes (tests/cpp/test_solver_lin.cpp, mode
    "timed") reads the exported system from a file, repeats setMatrix / solveProblem and reports the median wall time per
    call split into ingress (host CSR -> device, PCIe inclusive) / preconditioner set-up / Krylov (+ b, x transfers)."""
    import subprocess
    import tempfile
    from isph_amd import build
    exe = build.build_cpp_test()
    rp, ci, val = A.export_csr()
    bh = b.cpu().numpy()
    n = len(rp) - 1
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=base) as td_:
        fin, fout = os.path.join(td_, "sys.bin"), os.path.join(td_, "x.bin")
        with open(fin, "wb") as f:
            np.array([n, len(val)], np.int32).tofile(f)
            rp.astype(np.int32).tofile(f); ci.astype(np.int32).tofile(f)
            val.tofile(f); bh.tofile(f)
        cmd = [exe, fin, fout, "1", "timed", str(repeat), str(int(sub_rows))]
        if coords is not None:                      # PrecondWrapper_Ifpack::setCoordinates: x[n], y[n], z[n]
            fc = os.path.join(td_, "coords.bin")
            np.ascontiguousarray(np.asarray(coords)[:n, :3].T).tofile(fc)
            cmd.append(fc)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": (r.stdout + r.stderr)[-400:]}
        rec = None
        for line in r.stdout.splitlines():
            if line.startswith('{"dropin"'):
                rec = json.loads(line)["dropin"]
            if line.startswith('{"ingress_last_call"') and rec is not None:
                rec["ingress_last_call"] = json.loads(line)["ingress_last_call"]
        if rec is None:
            return {"error": "no record from the driver: " + r.stdout[-300:]}
        xd = np.fromfile(fout)[:n]
    rec["path"] = ("SolverLin_Belos::solveProblem(PrecondWrapper_Ifpack) on a host CSR in the caller's atom order, C++ mirror, median of %d calls%s"
                   % (repeat, "; the adapter called PrecondWrapper_Ifpack::setCoordinates (INTEGRATION.md): rows numbered by the library" if coords is not None
                      else "; pair_isph.cpp unchanged: subdomains = 512 consecutive rows of the atom order" if not sub_rows else ""))
    return rec, xd


def step_workload(args, json_fd=None, ctx=None, steps=None, warmup=None, quiet=False):
    """The reference's benchmark protocol as a measured record: `run 20` of
    bench-script/hopper/tgv/1728/tgv-3d-p24.lmp with tgv.xml -- 3-D TGV on a simple-cubic lattice, Quintic kernel cut 3h
    (--kernel quintic; wendland = the sph-script), theta 0.5, "Singular Poisson" DoubleDiag, dt = h/8 (.lmp:130), fixes
    isph + isph/shift 0.05 -- on ONE GPU.  Every step runs PairISPH::compute as the reference does (pair_isph.cpp:1241-1380,
    SURVEY 3.1): computePre (volumes, G_i, L_i) -> Helmholtz assembly + 3-component solve -> Poisson assembly -> solve
    -> zero-mean pressure -> velocity / pressure correction, then advanceTime (fix isph) and the particle shift (fix
    isph/shift).  Matrices and preconditioners are rebuilt every step.  Between steps the host plays LAMMPS: wrap, ghost
    atoms, neighbour list (workload.make_cloud) -- timed and reported, not part of the path.
    value = steps / (sum of the device stages), the scope of the reference's `ISPH:` timers (utils.cpp:37-38)."""
    import torch
    import isph_amd  # noqa: F401
    from isph_amd import hip, workload
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    own_ctx = ctx is None
    if own_ctx:
        dev = torch.device("cuda", 0)
        tstream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(tstream)
        ctx = hip.Context(0, stream=tstream.cuda_stream, ordering="bricks" if args.library_order == "on" else "caller")
    dev = torch.device("cuda", torch.cuda.current_device())
    nsteps = args.steps if steps is None else steps
    nwarm = args.warmup if warmup is None else warmup
    lib_order = ctx.ordering == "bricks"
    ctx.set_periodic_box((0.0, 0.0, 0.0), (2 * np.pi,) * 3, (1, 1, 1))     # domain->boxlo / boxhi / periodicity of the script's box
    n = args.ncell
    # the atom order of the particles: the generator's bricks only when the caller's order is what the matrix keeps
    brick = tuple(int(t) for t in args.brick.split(",")) if (not lib_order or args.order == "bricks") else (n, n, n)
    coh = 2.0 if args.kernel == "wendland" else 3.0
    spec = workload.TGVSpec(dim=3, ncell=(n, n, n), brick=brick, mode=workload.LATTICE, kernel=args.kernel, cut_over_h=coh)
    parts0 = workload.make_tgv(spec)
    N = parts0["nlocal"]
    L = 2.0 * np.pi
    dt = 0.125 * spec.h                                   # timestep of the bench script (tgv-3d-p24.lmp:130 = h/8)
    brows = brick[0] * brick[1] * brick[2]
    bptr = np.arange(0, N + brows, brows).clip(0, N).astype(np.int32) if (not lib_order and all(n % k == 0 for k in brick) and brows <= 1024) else None
    smode = {"nullspace": hip.NULLSPACE, "pinzero": hip.PINZERO, "doublediag": hip.DOUBLEDIAG}[args.singular]
    x = torch.from_numpy(np.ascontiguousarray(parts0["x"][:N])).to(dev)
    v = torch.from_numpy(np.ascontiguousarray(parts0["v"][:N])).to(dev)
    pr = torch.zeros(N, dtype=torch.float64, device=dev)
    g = np.zeros(3)
    stages = ("neighbour_host", "upload", "computePre", "helmholtz_assemble", "helmholtz_solve", "poisson_assemble", "poisson_solve",
              "correct_advance", "shift")
    acc = {k: 0.0 for k in stages}
    its = {"helmholtz": [], "poisson": []}
    substats = []

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    def make_prec(A, singular):
        if args.prec == "sa-amg":                            # ml.xml: max levels 10, Gauss-Seidel 4 sweeps pre and post
            nv = torch.full((N,), 1.0 / np.sqrt(float(N)), dtype=torch.float64, device=dev) if singular else None
            # ml.xml: "ML Gauss-Seidel" with "efficient symmetric", 4 sweeps: forward before, backward after the coarse correction
            # (--amg-smoother symmetric: 4 symmetric sweeps on both sides, what rounds 2-4 ran); max levels: ml.xml asks for 10,
            # the library's hierarchy holds 8 (4 are reached at 10^6 rows)
            return hip.PrecondAMG(ctx, A, nullvec=nv, params=hip.AmgParams(max_levels=8, sweeps=4, block=args.block, theta=args.amg_theta,
                                                                             smoother=0 if args.amg_smoother == "symmetric" else 1))
        if args.prec.startswith("bjacobi-ilu") and lib_order:
            return hip.Precond(ctx, A, args.prec, 0)                       # the library's bricks
        if args.prec == "bjacobi-ilu0" and bptr is not None:
            return hip.Precond(ctx, A, args.prec, block_ptr=bptr)
        return hip.Precond(ctx, A, args.prec, args.block)

    total = nwarm + nsteps
    for step in range(total):
        timed = step >= nwarm
        t0 = time.perf_counter()
        cloud = workload.make_cloud(x.cpu().numpy(), (L, L, L), spec.h, spec.cut, like=parts0)
        t1 = time.perf_counter()
        own = torch.from_numpy(cloud["owner_index"].astype(np.int64)).to(dev)
        dp = dict(cloud)
        for k in ("x", "type", "neigh_ptr", "neigh_idx"):
            dp[k] = torch.from_numpy(np.ascontiguousarray(cloud[k])).to(dev)
        colmap = own.to(torch.int32).contiguous()
        rho = torch.from_numpy(cloud["rho"]).to(dev)
        nu = torch.from_numpy(cloud["nu"]).to(dev)
        vall = v[own].contiguous()
        pall = pr[own].contiguous()
        zeros3 = torch.zeros_like(vall)
        # the functors of one time step all walk the same neighbour list (LAMMPS rebuilds it between steps at most): the
        # library builds its layout of the list once per step instead of once per operator call
        ctx.hold_neighbours(True)
        t2 = sync()
        # ---- computePre (pair_isph_corrected.cpp:302-313): V_i, then G_i and L_i (always formed, whatever the family)
        vf = hip.compute_volumes(ctx, dp, colmap, kernel=args.kernel)
        vfrac = vf[own].contiguous()
        Gc, Lc = hip.compute_corrections(ctx, dp, colmap, vfrac, kernel=args.kernel)
        t3 = sync()
        # ---- Helmholtz (pair_isph.cpp:932-982)
        if args.theta < 1e-24:
            H, bh = hip.assemble_helmholtz(ctx, dp, colmap, dt, 0.0, nu, rho, pall, zeros3, g, vall, vfrac=vfrac, kernel=args.kernel, rhs_only=True)
            t4 = t5 = sync()
            xh = bh
        else:
            H, bh = hip.assemble_helmholtz(ctx, dp, colmap, dt, args.theta, nu, rho, pall, zeros3, g, vall, vfrac=vfrac, kernel=args.kernel)
            t4 = sync()
            MH = make_prec(H, False)
            xh = torch.cat([v[:, 0], v[:, 1], v[:, 2]]).contiguous()       # x = View(vstar): the current velocity
            ih = hip.solve(ctx, H, bh, xh, prec=MH, singular=False, nvec=3, lda=N)
            MH.close(); H.close()
            its["helmholtz"].append(int(ih.iters))
            t5 = sync()
        vstar = torch.stack([xh[:N], xh[N:2 * N], xh[2 * N:3 * N]], dim=1).contiguous()
        vstar_all = vstar[own].contiguous()
        # ---- Poisson (pair_isph.cpp:988-1023)
        A, b = hip.assemble_poisson(ctx, dp, colmap, dt, rho, vstar_all, vfrac=vfrac, kernel=args.kernel, singular=smode)
        t6 = sync()
        M = make_prec(A, smode == hip.NULLSPACE)
        dpv = torch.zeros(N, dtype=torch.float64, device=dev)
        ip = hip.solve(ctx, A, b, dpv, prec=M, singular=(smode == hip.NULLSPACE))
        M.close()
        sub = A.subdomains() if (lib_order and timed) else None           # (a 8 KB table; inside the Poisson bracket like the close)
        A.close()
        its["poisson"].append(int(ip.iters))
        dpv -= dpv.mean()                                                  # computeZeroMeanPressure
        t7 = sync()
        if sub is not None:
            sz = np.diff(sub)
            substats.append((int(len(sz)), int(sz.min()), float(sz.mean()), int(sz.max())))
        # ---- correction + advanceTime (pair_isph.cpp:1030-1031, pair_isph_corrected.cpp:1172-1199)
        dp_all = dpv[own].contiguous()
        hip.correct_velocity_pressure(ctx, dp, colmap, dt, rho, dp_all, vstar_all, pall, vfrac, kernel=args.kernel)
        dpa = hip.advance_begin(ctx, dp, colmap, dt, pall, vall, vstar_all, vfrac, kernel=args.kernel)
        xall = dp["x"]
        hip.advance_end(ctx, N, 3, dt, dpa, vstar_all, pall, xall, vall)
        t8 = sync()
        # ---- fix isph/shift 0.05 (fix_isph_shift.cpp:146-163): computePre on the moved particles, then shiftParticles
        vall2 = vall[:N][own].contiguous()
        pall2 = pall[:N][own].contiguous()
        dp["x"] = (xall[:N][own] + (xall - xall[own])).contiguous()        # ghosts follow their owners
        vf2 = hip.compute_volumes(ctx, dp, colmap, kernel=args.kernel)
        vfrac2 = vf2[own].contiguous()
        hip.shift_particles(ctx, dp, colmap, 0.05, spec.cut, 0.1, dt, dp["x"], vall2, pall2, vfrac2, kernel=args.kernel)
        t9 = sync()
        ctx.hold_neighbours(False)
        x, v, pr = dp["x"][:N].clone(), vall2[:N].clone(), pall2[:N].clone()
        if timed:
            for k, d in zip(stages, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8)):
                acc[k] += d
        if not quiet:
          sys.stderr.write("step %d: pre %.1f  helm asm %.1f solve %.1f [%s its]  poisson asm %.1f solve %.1f [%d its]  corr+adv %.1f  shift %.1f  | host neighbours %.0f ms\n"
                         % (step, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, its["helmholtz"][-1] if its["helmholtz"] else "-",
                            (t6 - t5) * 1e3, (t7 - t6) * 1e3, ip.iters, (t8 - t7) * 1e3, (t9 - t8) * 1e3, (t1 - t0) * 1e3))
    K = nsteps
    dev_s = sum(acc[k] for k in stages[2:])
    # sanity of the physics: the kinetic energy of the decaying vortex only goes down, the velocity stays finite
    ke = float((v * v).sum().item()) * 0.5
    out = {
        "metric": "ISPH time steps/sec (3D TGV, %d^3 particles, %s cut %gh, theta %g, %s, GMRES(50)+%s; PairISPH::compute + fix isph + fix isph/shift per step)"
                  % (n, args.kernel, coh, args.theta, args.singular, args.prec),
        "value": K / dev_s, "unit": "steps/s", "n_gpus": 1, "steps": K, "warmup": nwarm,
        "ms_per_step": dev_s / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "reference benchmark protocol bench-script/hopper/tgv/1728 (tgv.xml, tgv-3d-p24.lmp: run 20), one GPU",
                   "rows": N, "kernel": args.kernel, "cut_over_h": coh, "theta": args.theta, "singular": args.singular, "precond": args.prec,
                   "dt": dt, "library_row_order": "bricks" if lib_order else "caller",
                   "subdomains": "the library's bricks" if lib_order else (("%dx%dx%d bricks" % brick) if bptr is not None else "%d rows" % args.block),
                   "iterations_poisson": its["poisson"][nwarm:], "iterations_helmholtz_3rhs_total": its["helmholtz"][nwarm:],
                   "subdomains_per_step_count_min_mean_max": substats,
                   "kinetic_energy_sum_end": ke},
        "stages_ms_per_step": {k: acc[k] / K * 1e3 for k in stages},
        "split": {"ISPH: computePre": acc["computePre"] / K * 1e3,
                  "ISPH: computeHelmholtz": acc["helmholtz_assemble"] / K * 1e3, "ISPH: solveHelmholtz": acc["helmholtz_solve"] / K * 1e3,
                  "ISPH: computePoisson": acc["poisson_assemble"] / K * 1e3, "ISPH: solvePoisson": acc["poisson_solve"] / K * 1e3,
                  "assembly_ms": (acc["computePre"] + acc["helmholtz_assemble"] + acc["poisson_assemble"]) / K * 1e3,
                  "solve_ms": (acc["helmholtz_solve"] + acc["poisson_solve"]) / K * 1e3},
        "host_lammps_side_ms_per_step": {"neighbour_list_and_ghosts": acc["neighbour_host"] / K * 1e3, "upload": acc["upload"] / K * 1e3,
                                         "note": "what LAMMPS does between compute() calls; not in `value`"},
    }
    if json_fd is not None:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if own_ctx:
        ctx.close()
    return out


def self_launch(args):
    """`python bench.py --gpus N` without torch.distributed.run: this process never touches the GPU; it starts one
    fresh child per rank (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* like torchrun), waits, and exits non-zero if a rank failed.
    Rank 0's JSON line goes straight to our stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live:                                   # a rank that dies must not leave the others waiting in a collective
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
                for q in live:
                    q.kill()                      # exact children of this process, by handle
    sys.exit(rc)


class Case:
    """The resident system of one run: this rank's particles in one atom order, the assembled matrix (twice: the first
    pass pays the one-time device allocations, the second is the steady state a time step sees), and step() = one pass of
    `ISPH: solvePoisson` (preconditioner build + FGMRES + projections)."""

    def __init__(self, env, order, lib_order):
        import torch
        from isph_amd import hip, dist
        self.env, self.order, self.lib_order = env, order, lib_order
        ctx, dev, args, td, world, rank = (env[k] for k in ("ctx", "dev", "args", "td", "world", "rank"))
        ctx.set_ordering("bricks" if lib_order else "caller")
        # the periodic box, as a LAMMPS adapter knows it (domain->boxlo / boxhi / periodicity): periodic along the axes this
        # rank's brick spans alone
        pgb = pgrid_for(world)
        ctx.set_periodic_box((0.0, 0.0, 0.0), (2 * np.pi,) * 3, tuple(int(g == 1) for g in pgb))
        t0 = time.perf_counter()
        self.spec, parts, self.brick = make_particles(args, world, rank, order)
        if world > 1:
            parts = dist.prune_ghosts(parts)                  # ghost columns = the referenced tags only (Epetra's column map)
        if args.force_rccl and world == 1:
            plan = dist.make_self_halo_plan(parts)
        else:
            plan = dist.make_plan(parts, td)                  # column map + halo lists (trivial on 1 rank)
        self.parts, self.plan = parts, plan
        self.host_generate_s = time.perf_counter() - t0
        nlocal = self.nlocal = parts["nlocal"]
        n, brick = args.ncell, self.brick
        # rows in the caller's order: block-Jacobi subdomains = the generator's bricks when they tile this rank's lattice
        brows = brick[0] * brick[1] * brick[2]
        self.bptr = None
        if (not lib_order and order == "bricks" and all(n % k == 0 for k in brick) and brows <= 1024 and args.prec == "bjacobi-ilu0"):
            assert nlocal % brows == 0, "the generator's bricks must tile the rank's rows"
            self.bptr = np.arange(0, nlocal + brows, brows).clip(0, nlocal).astype(np.int32)
        dparts = dict(parts)
        for k in ("x", "type", "neigh_ptr", "neigh_idx"):
            dparts[k] = torch.from_numpy(np.ascontiguousarray(parts[k])).to(dev)
        self.dparts = dparts
        self.colmap = torch.from_numpy(plan.colmap).to(dev)
        self.rho = torch.from_numpy(parts["rho"]).to(dev)
        self.vstar = torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)
        self.own = torch.from_numpy(parts["owner_index"].astype(np.int64)).to(dev)
        # computePre (volumes on the GPU + forward comm of ghost volumes) and the Poisson assembly
        self.fwd = hip.HaloForward(ctx, nlocal, plan.peers, plan.send_ptr, plan.send_idx, plan.recv_ptr) if plan.npeers else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.A, self.b = self.assemble()
        torch.cuda.synchronize()
        self.assemble_first_ms = (time.perf_counter() - t0) * 1e3
        self.A.close()
        t0 = time.perf_counter()
        self.A, self.b = self.assemble()
        torch.cuda.synchronize()
        self.assemble_ms = (time.perf_counter() - t0) * 1e3
        # the same once more with a synchronisation between computePre's volumes and the matrix assembly, for the split
        self.A.close()
        self.A, self.b = self.assemble(split=True)
        self.x = torch.zeros(nlocal, dtype=torch.float64, device=dev)
        self.bwork = torch.empty_like(self.b)
        self.prm = hip.SolverParams()
        self.pinfo = {}
        self.nullvec = torch.full((nlocal,), 1.0 / np.sqrt(float(nlocal * world)), dtype=torch.float64, device=dev)

    def assemble(self, split=False):
        import torch
        from isph_amd import hip, dist
        env, args = self.env, self.env["args"]
        ctx = env["ctx"]
        if split:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        vf = hip.compute_volumes(ctx, self.dparts, self.colmap, kernel=args.kernel)
        if self.fwd is None:
            vfrac = vf[self.own].contiguous()
        else:
            vfrac = dist.forward_scalar_rccl(self.fwd, self.plan, vf)   # forward_comm_pair of Vfrac over the library's RCCL comm
        if split:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            A, b = self._assemble_matrix(vfrac)
            torch.cuda.synchronize()
            self.assemble_split_ms = {"compute_volumes_and_forward": (t1 - t0) * 1e3, "assemble_poisson": (time.perf_counter() - t1) * 1e3}
            return A, b
        return self._assemble_matrix(vfrac)

    def _assemble_matrix(self, vfrac):
        from isph_amd import hip
        env, args = self.env, self.env["args"]
        ctx = env["ctx"]
        A, b = hip.assemble_poisson(ctx, self.dparts, self.colmap, self.spec.dt, self.rho, self.vstar, vfrac=vfrac,
                                    ncol=self.plan.ncol, kernel=args.kernel, rank0=(env["rank"] == 0))
        if self.plan.npeers:
            A.set_halo(self.plan.peers, self.plan.send_ptr, self.plan.send_idx, self.plan.recv_ptr)
        return A, b

    def subdomains(self):
        """what the block-Jacobi preconditioner of this case is built on, for the record"""
        if self.lib_order:
            o = self.A.ordering()
            sz = np.diff(o["block_ptr"])
            g = o["geom"]
            return dict(kind="the library's bricks (isph_ctx_set_ordering BRICKS): %dx%dx%d cells of the mean spacing" % tuple(g.cells_per_brick),
                        count=int(len(sz)), rows_min=int(sz.min()), rows_max=int(sz.max()), rows_mean=float(sz.mean()))
        if self.bptr is not None:
            return dict(kind="%dx%dx%d-cell bricks of the generator's numbering (isph_prec_create_blocks)" % self.brick,
                        count=int(len(self.bptr) - 1), rows_min=int(np.diff(self.bptr).min()), rows_max=int(np.diff(self.bptr).max()))
        return dict(kind="%d consecutive rows of the caller's atom order" % self.env["args"].block)

    def make_prec(self, prec):
        from isph_amd import hip, dist
        env, args, A, pinfo = self.env, self.env["args"], self.A, self.pinfo
        ctx = env["ctx"]
        if prec == "sa-amg":
            M = hip.PrecondAMG(ctx, A, nullvec=self.nullvec, params=hip.AmgParams(block=args.block, theta=args.amg_theta))
            if not pinfo:
                pinfo.update(levels=[M.level_info(l) for l in range(M.levels)])
        elif prec.startswith("schwarz-ilu"):
            M = hip.PrecondSchwarz(ctx, A, level_of_fill=int(prec[-1]), overlap=args.overlap, combine=args.combine,
                                   block_size=args.block)
            if not pinfo:
                pinfo.update(M.schwarz_info())
            pinfo["create_ms"] = M.create_timing()          # of the last create
        elif prec in ("ilu0", "ilu1"):
            M = hip.PrecondSchwarz(ctx, A, level_of_fill=int(prec[-1]), overlap=0, block_size=0)
            if not pinfo:
                pinfo.update(M.schwarz_info())
        elif prec.startswith("overlap-ilu"):
            # Ifpack on N ranks with "Overlap Level" 1: this rank's rows + the rows of its ghost columns, one ILU(k) block
            plan = self.plan
            assert plan.npeers, "overlap-ilu<k> needs ghost columns (--gpus > 1 or --force-rccl)"
            rpl, cil, vall = A.export_csr()
            rpe, cie, ve = dist.extend_rows(plan, rpl, cil, vall, env["td"])
            Aext = hip.Matrix.from_csr(ctx, rpe, cie, ve)
            M = hip.PrecondOverlap(ctx, Aext, plan, level_of_fill=int(prec[-1]), combine=args.combine)
            Aext.close()
            if not pinfo:
                pinfo.update(extended_rows=len(rpe) - 1, extended_nnz=int(rpe[-1]))
        elif prec.startswith("bjacobi-ilu") and self.lib_order:
            M = hip.Precond(ctx, A, prec, 0)                # the library's bricks
        elif prec == "bjacobi-ilu0" and self.bptr is not None:
            M = hip.Precond(ctx, A, prec, block_ptr=self.bptr)
        else:
            M = hip.Precond(ctx, A, prec, args.block)
        if not pinfo and prec == args.prec and prec.startswith("bjacobi-ilu"):
            pinfo.update(M.info(), factor_nnz_exact=int(M.info()["factor_nnz"]))
        return M

    def step(self, prec=None):
        from isph_amd import hip
        prec = prec or self.env["args"].prec
        self.bwork.copy_(self.b)
        self.x.zero_()
        M = self.make_prec(prec)
        inf = hip.solve(self.env["ctx"], self.A, self.bwork, self.x, prec=M, singular=True, params=self.prm)
        M.close()
        return inf

    def timed(self, steps, warmup, prec=None):
        """ms per step() over `steps` after `warmup` (single rank: no barrier)"""
        import torch
        for _ in range(warmup):
            inf = self.step(prec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            inf = self.step(prec)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, inf

    def close(self):
        self.A.close()
        if self.fwd is not None:
            self.fwd.close()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        self_launch(args)
    # stdout carries exactly ONE line, the JSON record.  Libraries write banners to the C-level stdout (RCCL prints its
    # version block there when a communicator is created), so fd 1 is pointed at stderr for the whole run and the record
    # is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.workload == "step":
        assert args.gpus == 1, "--workload step runs on one GPU"
        return step_workload(args, json_fd)
    import torch
    import isph_amd  # noqa: F401
    from isph_amd import hip, workload, dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ngpu = args.gpus
    assert world == ngpu, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (ngpu, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one non-default stream for torch's own kernels AND the library: every producer/consumer pair is stream-ordered
    # (handing the library the legacy default stream would make it create a private non-blocking one)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    transport = None
    if world > 1 and args.share_gpu:
        import torch.distributed as td
        td.init_process_group("gloo")
        transport = dist.td_host_transport(td)            # kept alive until the context is gone
        ctx = hip.Context(0, stream=stream, rank=rank, nranks=world, transport=transport)
    elif world > 1:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=dev)
        uid = [hip.Context.unique_id() if rank == 0 else None]
        td.broadcast_object_list(uid, src=0)
        ctx = hip.Context(local_rank, stream=stream, rank=rank, nranks=world, uid=uid[0])
    elif args.force_rccl:
        td = None
        ctx = hip.Context(local_rank, stream=stream, rank=0, nranks=1, uid=hip.Context.unique_id())
    else:
        td = None
        ctx = hip.Context(local_rank, stream=stream)

    lib_order = args.library_order == "on"
    if args.prec.startswith("overlap-ilu"):
        lib_order = False                                 # the extended matrix is built from the exported rows (caller's numbering)
    ctx.set_ordering("bricks" if lib_order else "caller")
    env = dict(ctx=ctx, dev=dev, args=args, td=td, world=world, rank=rank)
    case = Case(env, args.order, lib_order)
    spec, parts, plan, nlocal, brick, bptr = case.spec, case.parts, case.plan, case.nlocal, case.brick, case.bptr
    brows = brick[0] * brick[1] * brick[2]
    n = args.ncell
    pg = pgrid_for(world)
    A, b = case.A, case.b
    assemble_first_ms, assemble_ms = case.assemble_first_ms, case.assemble_ms
    info_m = A.info()
    x, prm, pinfo = case.x, case.prm, case.pinfo
    step = case.step

    def barrier():
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        inf = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        inf = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # the per-launch SpMV duration (HIP events on the library's stream around every in-solve SpMV) is taken in
    # separate, untimed passes so that the event records do not sit inside the timed steps
    ctx.set_profile(True)
    spmv_ms, spmv_calls = 0.0, 0
    nprof = 2
    for _ in range(nprof):
        infp = step()
        spmv_ms += infp.spmv_ms
        spmv_calls += infp.spmv_calls
    barrier()
    prof = ctx.profile_read()          # {class: (ms, launches)} over the nprof untimed passes (set-up + solve)
    halo_prof = ctx.halo_profile_read() if world > 1 else None
    x_headline = x.clone()             # the headline configuration's solution (the alt runs below reuse x)
    if td is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    # isolated SpMV (no halo, back-to-back launches) for reference
    ctx.set_profile(False)
    xin = torch.randn(plan.ncol, dtype=torch.float64, device=dev)
    yout = torch.empty(nlocal, dtype=torch.float64, device=dev)
    iso_ms = A.spmv_time(xin, yout, reps=args.spmv_reps) if (world == 1 and not args.force_rccl) else None

    alg_bytes = spmv_algorithmic_bytes(info_m["nrow"], info_m["nnz"])
    # HBM traffic of the SpMV kernel from the PMC passes committed under profiles/ (rocprofv3
    # cannot run inside this process); only quoted when it was taken on this very matrix.
    traffic = None
    import glob
    tsrc = None
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_spmv_traffic.json")), reverse=True):  # newest round first
        tj = json.load(open(tpath))
        if tj.get("nrow") == info_m["nrow"] and tj.get("nnz") == info_m["nnz"]:
            traffic, tsrc = tj["traffic_bytes_per_launch"], os.path.relpath(tpath, ROOT)
            break
    avg_ms = spmv_ms / max(spmv_calls, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # what the production kernel (16-bit window columns) is built to move: 10 B per STORED entry (padding included),
    # the per-slice window table, x once, y once
    nslices = (info_m["nrow"] + 63) // 64
    moved_model = 10 * info_m["stored"] + 64 * 4 * nslices + 16 * info_m["nrow"]

    # ---- N > 1: a record that validates itself (every rank takes part; rank 0 prints)
    multi = None
    if world > 1:
        ctx.set_profile(False)
        cpu_side = args.share_gpu                      # gloo reduces host tensors
        red = lambda t, op: (td.all_reduce(t, op=op), t)[1]
        tdev = "cpu" if cpu_side else dev
        # explicit residual of the GLOBAL system from the distributed product: b (projected by the solve) - A x, with the
        # halo exchange of the production SpMV; A 1 = 0 across every rank boundary; the scale of a row from A s, s = +-1
        r = case.bwork - A.spmv(x_headline)
        sums = red(torch.stack([(r * r).sum(), (case.bwork * case.bwork).sum(), x_headline.sum(), r.sum()]).to(tdev), td.ReduceOp.SUM)
        # the operator of the singular solve is y = A x - (A x . n) n (PoissonProjection::Apply, solver_lin.h:131-140): the
        # residual that converges is the one with its component along n = 1/sqrt(N) removed
        rr_proj = max(sums[0].item() - sums[3].item() ** 2 / float(nlocal * world), 0.0)
        ones = torch.ones(nlocal, dtype=torch.float64, device=dev)
        sgn = torch.where(torch.rand(nlocal, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + rank)) < 0.5, -1.0, 1.0).to(torch.float64)
        mx = red(torch.stack([A.spmv(ones).abs().max(), A.spmv(sgn).abs().max()]).to(tdev), td.ReduceOp.MAX)
        mine = dict(rank=rank, comm=ctx.comm_info(), peers=int(plan.npeers), ghost_cols=int(plan.ncol - nlocal),
                    halo_bytes_per_spmv=int(8 * (int(plan.send_ptr[-1]) + int(plan.recv_ptr[-1]))) if plan.npeers else 0,
                    rows=int(nlocal), iterations=int(inf.iters), converged=int(inf.converged), subdomains=case.subdomains().get("count"),
                    halo_profile=halo_prof and dict(products=halo_prof["products"],
                                                    exchange_us=halo_prof["exchange_ms"] / max(halo_prof["products"], 1) * 1e3,
                                                    interior_us=halo_prof["interior_ms"] / max(halo_prof["products"], 1) * 1e3,
                                                    exposed_us=halo_prof["exposed_ms"] / max(halo_prof["products"], 1) * 1e3))
        allr = [None] * world
        td.all_gather_object(allr, mine)
        its_all = [q["iterations"] for q in allr]
        assert len(set(its_all)) == 1, "ranks report different iteration counts: %s" % its_all
        assert all(q["comm"]["ranks"] == world for q in allr), "a rank's communicator does not span the job: %s" % [q["comm"] for q in allr]
        multi = dict(rccl_ranks=int(allr[0]["comm"]["ranks"]), transport=allr[0]["comm"]["transport"], per_rank=allr,
                     iterations_equal_on_all_ranks=True,
                     global_rel_residual=float(np.sqrt(rr_proj / sums[1].item())),
                     global_rel_residual_unprojected_operator=float(np.sqrt(sums[0].item() / sums[1].item())),
                     global_solution_sum_over_abs_scale=float(abs(sums[2].item())),
                     a_times_one_max_abs=float(mx[0].item()), a_times_signs_max_abs=float(mx[1].item()),
                     note="global_rel_residual = ||P (b - A x)|| / ||b|| of the global system (b projected, P the null-space projection of "
                          "the solve), from the distributed product, all-reduced over the ranks; ..._unprojected_operator = the figure "
                          "solver_lin_belos.h:201-212 prints on failure (A without the projection); A 1 = 0 holds across rank boundaries when a_times_one_max_abs is round-off of "
                          "a_times_signs_max_abs; halo_profile: HIP events per product in the profile passes")
        assert multi["global_rel_residual"] < 1e-6, multi
        assert multi["a_times_one_max_abs"] <= 1e-9 * multi["a_times_signs_max_abs"], multi
        # the same brick alone on this process' GPU, periodic: what N = 1 gives here, for the agreement check of the curve
        if rank == 0 and not args.share_gpu:
            ctx1 = hip.Context(local_rank, stream=stream, ordering="bricks" if lib_order else "caller")
            c1 = Case(dict(ctx=ctx1, dev=dev, args=args, td=None, world=1, rank=0), args.order, lib_order)
            ms1, i1 = c1.timed(3, 2)
            multi["n1_in_process"] = dict(ms_per_step=ms1, iterations=int(i1.iters), solves_per_s=1e3 / ms1,
                                          note="rank 0's GPU alone, one periodic %d^3 brick, 3 steps after 2" % n)
            c1.close(); ctx1.close()
        td.barrier()

    if rank == 0:
        out = {
            "metric": METRIC_NAME.get(args.prec, "pressure-Poisson solves/sec (3D TGV, 1M particles per GPU, GMRES(50)+%s, tol 1e-8)" % args.prec),
            "value": args.steps * world / elapsed,
            "unit": "solves/s (1M-particle bricks; x n_gpus under weak scaling)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D Taylor-Green vortex, %d^3 particles per GPU, Wendland cut=2h (BASELINE configs[1]%s), "
                                   "state after one Lagrangian step (mode=%s)" % (n, "" if world == 1 else "/[2] brick", args.mode),
                       "rows_per_gpu": nlocal, "global_rows": nlocal * world, "nnz_per_gpu": info_m["nnz"],
                       "nnz_per_row": info_m["nnz"] / max(nlocal, 1), "sell_padding": info_m["stored"] / max(info_m["nnz"], 1),
                       "solver": "FGMRES(50) DGKS tol 1e-8, right prec", "precond": args.prec,
                       "atom_order": args.order, "library_row_order": "bricks" if lib_order else "caller",
                       "subdomains": case.subdomains(),
                       "parallelism": ("domain bricks %dx%dx%d, " % pg) +
                                      ("host-staged halo + all-reduce over gloo, ALL RANKS ON ONE GPU (rehearsal, not a performance figure)"
                                       if args.share_gpu else "RCCL halo + all-reduce"),
                       "iterations": inf.iters, "restarts": inf.restarts, "converged": inf.converged,
                       "rel_res": inf.rel_res_implicit, "assemble_ms": assemble_ms, "assemble_first_call_ms": assemble_first_ms,
                       "assemble_split_ms": case.assemble_split_ms,
                       "spmv_isolated_ms": iso_ms, ("amg" if args.prec == "sa-amg" else "ilu"): pinfo},
            "roofline": {"bound": "hbm", "kernel": "k_sell_spmv16<8,false> (SELL-64, 16-bit window columns)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_moved_model": moved_model,
                         "achieved_moved_model_GBs": moved_model / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                         "traffic_source": (tsrc + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)") if traffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms, "launches": spmv_calls},
        }
        out["roofline"].update(solve_roofline(info_m, pinfo, inf, prm, nprof, prof, elapsed / args.steps, args.prec))
        if world == 1 and not args.no_alt and args.prec == "bjacobi-ilu0":
            # the two other preconditioners of the C ABI on the SAME resident system, measured in this run (same step(): the
            # preconditioner is rebuilt every solve): point Jacobi and SA-AMG (PrecondWrapper_ML's stand-in)
            alt = {}
            for name in ("jacobi", "sa-amg"):
                for _ in range(2):
                    ia = step(name)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    ia = step(name)
                torch.cuda.synchronize()
                alt[name] = {"ms_per_solve": (time.perf_counter() - t0) / 3 * 1e3, "iterations": ia.iters, "converged": ia.converged,
                             "rel_res": ia.rel_res_implicit}
            out["alt"] = alt
        if world == 1 and not args.no_dropin and not args.force_rccl and args.prec == "bjacobi-ilu0" and (args.block == 512 or bptr is not None):
            # the unchanged SolverLin drop-in (host CSR in, host x out) beside the device-resident figure above
            # with the three-line adapter call that hands the wrapper the coordinates (the library numbers the rows), and
            # without it (pair_isph.cpp unchanged: the rows keep the atom order, 512 consecutive rows per subdomain)
            for key, kw in (("dropin", dict(coords=parts["x"]) if lib_order else dict(sub_rows=brows if bptr is not None else 0)),
                            ("dropin_unchanged_adapter", dict(sub_rows=0))):
                if key == "dropin_unchanged_adapter" and not lib_order:
                    continue
                d = dropin_leg(A, b, **kw)
                if isinstance(d, tuple):
                    rec, xd = d
                    xr = x_headline.cpu().numpy()
                    rec["x_rel_diff_vs_device_resident"] = float(np.linalg.norm(xd - xr) / np.linalg.norm(xr))
                    rec["iterations_device_resident"] = inf.iters
                    rec["ratio_to_device_resident"] = rec["ms_per_solve"] / (elapsed / args.steps * 1e3)
                    out[key] = rec
                else:
                    out[key] = d
        if world == 1 and not args.no_cpu_baseline and args.prec in ("none", "jacobi", "bjacobi-ilu0", "bjacobi-ilu1", "bjacobi-ilu2", "sa-amg"):
            rp, ci, val = A.export_csr()                                           # the caller's numbering
            bh = b.cpu().numpy()
            o = A.ordering()
            cb_bptr = bptr
            if o is not None:
                # the oracle gets the library's permutation and subdomain table explicitly: it solves P A P^T (P x) = P b
                import order as oorder
                rp, ci, val, bh = oorder.permute_system(rp, ci, val, bh, o["perm"])
                cb_bptr = o["block_ptr"] if args.prec.startswith("bjacobi-ilu") else None
            xkeep = []
            cb = cpu_baseline(rp, ci, val, bh, args.block, args.prec, args.amg_theta, args.cpu_ifpack_1rank, cb_bptr, keep_x=xkeep)
            # parity at BASELINE size, in the record: the oracle's solution of the same system with the same subdomains
            xc = xkeep[0]
            if o is not None:
                xu = np.empty_like(xc)
                xu[o["perm"]] = xc
                xc = xu
            xg = x_headline.cpu().numpy()
            cb["x_rel_diff_vs_gpu"] = float(np.linalg.norm(xg - xc) / np.linalg.norm(xc))
            cb["iterations_gpu"] = int(inf.iters)
            cb["iterations_equal"] = bool(int(inf.iters) == cb["same_blocks"]["iterations"])
            cb["parity_note"] = ("same_blocks = the oracle (CPU restatement of Belos FGMRES + Ifpack ILU(0)) on the exported system"
                                 + (", permuted with the library's row order and factored on the library's subdomain table" if o is not None else "")
                                 + "; tolerance of the parity tests: iterations +-1, x <= 1e-6")
            out["cpu_baseline"] = cb
            out["config"]["speedup_vs_cpu"] = out["value"] / cb["value"]            # against the FASTEST CPU variant
            for k in ("same_blocks", "block_per_core", "block_per_core_ilu1", "single_thread", "ifpack_1rank"):
                if k in cb:
                    out["config"]["speedup_vs_cpu_" + k] = out["value"] * cb[k]["seconds_per_solve"]
            if args.cpu_ifpack_1rank and "ifpack_1rank" in cb:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                json.dump(dict(nrow=info_m["nrow"], nnz=info_m["nnz"], ifpack_1rank=cb["ifpack_1rank"]),
                          open(os.path.join(ROOT, "gpurun_out", "r02_cpu_ifpack_1rank.json"), "w"))
        if multi is not None:
            out["multi_gpu"] = multi
        extra = {}
        if world == 1 and not args.no_orders and args.prec == "bjacobi-ilu0" and not args.force_rccl:
            # the same particles handed over in the other atom orders, with the library's own row numbering and without:
            # the headline must not depend on the order the caller's atoms happen to be in
            table = {}
            head_key = "%s/%s" % (args.order, "library" if lib_order else "caller")
            table[head_key] = dict(ms_per_solve=elapsed / args.steps * 1e3, iterations=int(inf.iters), converged=int(inf.converged),
                                   assemble_ms=assemble_ms, assemble_split_ms=case.assemble_split_ms, subdomains=case.subdomains(), headline=True)
            for order in ("lexicographic", "sortbin", "shuffled", "bricks"):
                for lib in (True, False):
                    key = "%s/%s" % (order, "library" if lib else "caller")
                    if key in table:
                        continue
                    c = Case(env, order, lib)
                    ms, ia = c.timed(3, 2)
                    table[key] = dict(ms_per_solve=ms, iterations=int(ia.iters), converged=int(ia.converged), assemble_ms=c.assemble_ms,
                                      assemble_split_ms=c.assemble_split_ms, subdomains=c.subdomains())
                    c.close()
            ctx.set_ordering("bricks" if lib_order else "caller")
            extra["atom_orders"] = dict(note="<atom order handed over>/<who numbers the matrix rows>; 3 timed solves after 2, same "
                                             "particles, preconditioner rebuilt every solve", cases=table)
        if world == 1 and not args.no_step and args.prec == "bjacobi-ilu0" and not args.force_rccl and args.kernel == "wendland":
            # the reference's own protocol beside the solve: 3 ISPH time steps, everything rebuilt per step
            st = step_workload(args, None, ctx=ctx, steps=3, warmup=1, quiet=True)
            extra["step"] = dict(metric=st["metric"], steps_per_s=st["value"], ms_per_step=st["ms_per_step"], split=st["split"],
                                 iterations_poisson=st["config"]["iterations_poisson"],
                                 iterations_helmholtz_3rhs_total=st["config"]["iterations_helmholtz_3rhs_total"],
                                 host_lammps_side_ms_per_step=st["host_lammps_side_ms_per_step"])
        if extra:
            out["extra"] = extra
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if td is not None:
        td.barrier()
        td.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
