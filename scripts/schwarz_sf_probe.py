"""Whole-matrix ILU(k) through the synchronisation-free sweeps, timed step by step (progress lines go to stdout at once)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import isph_amd
from isph_amd import hip, workload
import torch

def P(*a):
    print(*a, flush=True)

ns = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 16]
fill = int(sys.argv[2]) if len(sys.argv) > 2 else 0
both = len(sys.argv) <= 3 or sys.argv[3] != "sf"
brick = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ctx = hip.Context(0)
for n in ns:
    sp = workload.TGVSpec(dim=3, ncell=(n, n, n), brick=(brick,) * 3, mode=workload.ADVECT)
    p = workload.make_tgv(sp)
    colmap = workload.single_rank_colmap(p)
    vf = hip.compute_volumes(ctx, p, colmap)
    A, b = hip.assemble_poisson(ctx, p, colmap, sp.dt, p["rho"], np.ascontiguousarray(p["v"]),
                                vfrac=np.ascontiguousarray(vf[p["owner_index"]]))
    N = p["nlocal"]
    P("n", n, "rows", N, "brick", brick)
    t0 = time.perf_counter()
    Ms = hip.PrecondSchwarz(ctx, A, level_of_fill=fill, overlap=0, block_size=0)
    torch.cuda.synchronize()
    P("  create sync-free %.1f ms" % ((time.perf_counter() - t0) * 1e3), Ms.schwarz_info())
    P("  create stages, ms:", {k: round(v, 1) for k, v in Ms.create_timing().items()})
    r = torch.from_numpy(np.random.default_rng(0).standard_normal(N)).cuda()
    z = torch.empty_like(r)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Ms.apply(r, z)
        torch.cuda.synchronize()
        P("  apply sync-free %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    if both:
        t0 = time.perf_counter()
        Ml = hip.PrecondSchwarz(ctx, A, level_of_fill=fill, overlap=0, block_size=0, level_launches=True)
        torch.cuda.synchronize()
        P("  create level launches %.1f ms" % ((time.perf_counter() - t0) * 1e3))
        z2 = torch.empty_like(r)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Ml.apply(r, z2)
        torch.cuda.synchronize()
        P("  apply level launches %.2f ms" % ((time.perf_counter() - t0) * 1e3), "equal bits:", bool(torch.equal(z, z2)), "max rel diff %.2e" % float((z - z2).abs().max() / z2.abs().max()),
          "factor equal:", all(np.array_equal(a, c) for a, c in zip(Ms.export(), Ml.export())))
        Ml.close()
    x = np.zeros(N)
    t0 = time.perf_counter()
    info = hip.solve(ctx, A, b.copy(), x, prec=Ms, singular=True)
    P("  solve: %d iterations, converged %d, %.1f ms" % (info.iters, info.converged, (time.perf_counter() - t0) * 1e3))
    Ms.close(); A.close()
