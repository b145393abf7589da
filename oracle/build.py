"""Builds the CPU oracle (test infrastructure) with gcc.  Not part of the product."""
import os
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
