"""The SolverLin drop-in at BASELINE configs[1] size, timing only: assembles the 100^3 system on the GPU, exports it
to the host and runs the C++ driver of the mirror classes in "timed" mode (tests/cpp/test_solver_lin.cpp).
    python scripts/dropin_100.py [ncell] [repeat]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import isph_amd  # noqa: E402,F401
from isph_amd import build, hip, workload  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rep = sys.argv[2] if len(sys.argv) > 2 else "7"
keep = len(sys.argv) > 3 and sys.argv[3] == "keep"   # leave the system file under /dev/shm for a profiler run
exe = build.build_cpp_test()
ctx = hip.Context(0)
sp = workload.TGVSpec(dim=3, ncell=(n, n, n), brick=(8, 8, 8), mode=workload.ADVECT)
p = workload.make_tgv(sp)
colmap = workload.single_rank_colmap(p)
vf = hip.compute_volumes(ctx, p, colmap)
A, b = hip.assemble_poisson(ctx, p, colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]),
                            vfrac=np.ascontiguousarray(vf[p["owner_index"]]))
rp, ci, v = A.export_csr()
A.close(); ctx.close()
fin, fout = "/dev/shm/isph_dropin_sys.bin", "/dev/shm/isph_dropin_x.bin"
with open(fin, "wb") as f:
    np.array([len(rp) - 1, len(v)], np.int32).tofile(f)
    rp.astype(np.int32).tofile(f); ci.astype(np.int32).tofile(f); v.tofile(f); b.tofile(f)
try:
    r = subprocess.run([exe, fin, fout, "1", "timed", rep], capture_output=True, text=True, timeout=600)
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("{")), r.stderr[-500:])
finally:
    for f_ in (fin, fout):
        if os.path.exists(f_) and not keep:
            os.remove(f_)
