"""Builds the CPU oracle (test infrastructure) with gcc.  Not part of the product."""
import os
import sys
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libisph_oracle.so")


def build_oracle(force=False):
    srcs = [os.path.join(HERE, f) for f in ("isph_oracle.c", "isph_amg_oracle.c", "isph_schwarz_oracle.c")]
    deps = srcs + [os.path.join(HERE, "isph_oracle.h")]
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)
    if stale:
        r = subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
                            "-o", LIB] + srcs + ["-lm"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stderr)
    return LIB


REF_SRC = "/root/reference/IMPLICIT-SPH"
REF_DIR = os.path.join(HERE, "_ref")
REF_LIB = os.path.join(REF_DIR, "libisph_refkernels.so")


def build_ref(force=False):
    """oracle/_ref: the part of the reference that compiles from its own sources with nothing but the standard library
    -- the three SPH kernel classes and the particle-kind filter -- built where the sources lie (never copied), for checking the oracle against the
    real code.  Only possible where /root/reference exists (the build container); elsewhere the prebuilt library is
    used if it travelled along.  Returns the path or None."""
    shim = os.path.join(HERE, "ref_kernels_shim.cpp")
    if not os.path.isdir(REF_SRC):
        return REF_LIB if os.path.exists(REF_LIB) else None
    deps = [shim] + [os.path.join(REF_SRC, h) for h in ("kernel.h", "kernel_wendland.h", "kernel_quintic.h", "kernel_cubic.h", "filter.h", "functor.h", "mirror.h")]
    stale = force or not os.path.exists(REF_LIB) or any(os.path.getmtime(d) > os.path.getmtime(REF_LIB) for d in deps)
    if stale:
        os.makedirs(REF_DIR, exist_ok=True)
        r = subprocess.run(["g++", "-O2", "-std=c++11", "-ffp-contract=off", "-fPIC", "-shared", "-I", REF_SRC, "-o", REF_LIB, shim],
                           capture_output=True, text=True)
        if r.returncode != 0:
            # a checker-only artefact built from untrusted public sources: its failure must not fail the product build;
            # the tests that use it skip (tests/test_oracle.py) and say why
            sys.stderr.write("oracle/_ref: reference kernel build failed, continuing without it:\n" + r.stderr[-2000:] + "\n")
            return None
    return REF_LIB
