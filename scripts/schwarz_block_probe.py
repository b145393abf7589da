"""Block Schwarz ILU(0) at 100^3: create and application time, default form against forced level launches.
usage: python scripts/schwarz_block_probe.py [block rows] [overlap]"""
import sys, time, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import isph_amd
from isph_amd import hip, workload
ctx = hip.Context(0)
n = 100
sp = workload.TGVSpec(dim=3, ncell=(n, n, n), brick=(8,) * 3, mode=workload.ADVECT)
p = workload.make_tgv(sp)
colmap = workload.single_rank_colmap(p)
vf = hip.compute_volumes(ctx, p, colmap)
A, b = hip.assemble_poisson(ctx, p, colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]), vfrac=np.ascontiguousarray(vf[p["owner_index"]]))
N = p["nlocal"]
block = int(sys.argv[1]) if len(sys.argv) > 1 else 512
overlap = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for ll in (False, True):
    t0 = time.perf_counter()
    M = hip.PrecondSchwarz(ctx, A, level_of_fill=0, overlap=overlap, block_size=block, combine="zero", level_launches=ll)
    torch.cuda.synchronize()
    print("level launches" if ll else "default form", "create %.1f ms" % ((time.perf_counter() - t0) * 1e3), M.schwarz_info(), {k: round(v, 1) for k, v in M.create_timing().items()}, flush=True)
    r = torch.from_numpy(np.random.default_rng(0).standard_normal(N)).cuda(); z = torch.empty_like(r)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); M.apply(r, z); torch.cuda.synchronize()
        print("   apply %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    M.close()
