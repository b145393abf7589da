"""In-tree builds: host library (g++) and HIP library (hipcc, gfx950).  No JIT
cache: the .so files land next to the sources so they travel to the GPU box
with the snapshot."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")

HOST_LIB = os.path.join(PKG, "libisph_host.so")
HIP_LIB = os.path.join(PKG, "libisph_hip.so")

HOST_SRCS = ["workload.cpp", "lammps_formats.cpp"]
HIP_SRCS = ["isph_capi.hip"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r


def _headers():
    hs = [os.path.join(INC, f) for f in os.listdir(INC)]
    hs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".cuh"))]
    return hs


def build_host(force=False):
    srcs = [os.path.join(CSRC, s) for s in HOST_SRCS]
    if force or _stale(HOST_LIB, srcs + _headers()):
        _run(["g++", "-O2", "-std=c++17", "-fopenmp", "-fPIC", "-shared", "-I", INC, "-o", HOST_LIB] + srcs)
    return HOST_LIB


def hipcc_path():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def build_hip(force=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SRCS]
    deps = srcs + _headers() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")]
    if force or _stale(HIP_LIB, deps):
        _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-I", INC, "-I", CSRC, "-o", HIP_LIB] + srcs + ["-L/opt/rocm/lib", "-lrccl"])
    return HIP_LIB


CPP_TEST = os.path.join(ROOT, "tests", "cpp", "test_solver_lin")


def build_cpp_test(force=False):
    """C++ driver of the SolverLin / PrecondWrapper mirror classes (host/*.h),
    linked against libisph_hip.so; used by the -m gpu tests."""
    src = os.path.join(ROOT, "tests", "cpp", "test_solver_lin.cpp")
    host = os.path.join(PKG, "host")
    deps = [src] + [os.path.join(host, f) for f in os.listdir(host)] + [os.path.join(INC, "isph_hip.h")]
    if force or _stale(CPP_TEST, deps):
        build_hip()
        _run(["g++", "-O2", "-std=c++17", "-I", INC, "-I", host, "-o", CPP_TEST, src,
              "-L", PKG, "-lisph_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return CPP_TEST


RANK_THREADS = os.path.join(ROOT, "tests", "cpp", "librank_threads.so")


def build_rank_threads(force=False):
    """TEST transport (tests/cpp/rank_threads.cpp): N ranks as threads of one process for isph_ctx_create_hostcomm;
    plain g++, no device code."""
    src = os.path.join(ROOT, "tests", "cpp", "rank_threads.cpp")
    if force or _stale(RANK_THREADS, [src, os.path.join(INC, "isph_hip.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", INC, "-o", RANK_THREADS, src])
    return RANK_THREADS


MPI_INC, MPI_LIB = "/opt/conda/include", "/opt/conda/lib"
MPIEXEC = "/opt/conda/bin/mpiexec"
CPP_TEST_MPI = os.path.join(ROOT, "tests", "cpp", "test_solver_lin_mpi")
CPP_MPI_HOST = os.path.join(ROOT, "tests", "cpp", "test_mpi_host")


def have_mpi():
    """The image ships MPICH under /opt/conda (its `mpicxx` wrapper is broken, plain g++ with -I/-L works)."""
    return (os.path.exists(os.path.join(MPI_INC, "mpi.h")) and os.path.exists(MPIEXEC)
            and os.path.exists(os.path.join(MPI_LIB, "libmpi.so")) and os.path.exists(os.path.join(MPI_LIB, "libmpi.so.12")))


def build_cpp_mpi(force=False):
    """The multi-rank build of the C++ mirror (-DISPH_HAVE_MPI): the same driver as build_cpp_test with the real
    communicator, and the host-only MPI test of the row import / null vector.  Returns (driver, host_test) or None
    when no MPI is installed."""
    if not have_mpi():
        return None
    host = os.path.join(PKG, "host")
    hdeps = [os.path.join(host, f) for f in os.listdir(host)] + [os.path.join(INC, "isph_hip.h")]
    flags = ["-O2", "-std=c++17", "-DISPH_HAVE_MPI", "-I", MPI_INC, "-I", INC, "-I", host]
    # conda's lib directory also holds an older libstdc++ that must not shadow the system's (libisph_hip / ROCm need the
    # newer one): link libmpi by path and let the loader find it and its two conda-only dependencies through a private
    # directory of symlinks
    mpilib = os.path.join(ROOT, "tests", "cpp", "mpilib")
    os.makedirs(mpilib, exist_ok=True)
    for so in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):
        link = os.path.join(mpilib, so)
        if os.path.exists(os.path.join(MPI_LIB, so)) and not os.path.lexists(link):
            os.symlink(os.path.join(MPI_LIB, so), link)
    libs = ["-L", PKG, "-lisph_hip", os.path.join(MPI_LIB, "libmpi.so"), "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib",
            "-Wl,-rpath," + mpilib]
    for target, src in ((CPP_TEST_MPI, "test_solver_lin.cpp"), (CPP_MPI_HOST, "test_mpi_host.cpp")):
        src = os.path.join(ROOT, "tests", "cpp", src)
        if force or _stale(target, [src] + hdeps):
            build_hip()
            _run(["g++"] + flags + ["-o", target, src] + libs)
    return CPP_TEST_MPI, CPP_MPI_HOST


def build_all(force=False):
    return build_host(force), build_hip(force), build_cpp_test(force), build_cpp_mpi(force), build_rank_threads(force)
