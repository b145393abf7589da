import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import isph_amd
    from isph_amd import hip, workload, dist
    dev = torch.device("cuda", 0)
    ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    spec = workload.TGVSpec(dim=3, ncell=(100, 100, 100), brick=(8, 8, 8), mode=workload.ADVECT)
    parts = workload.make_tgv(spec)
    plan = dist.make_plan(parts, None)
    dp = dict(parts)
    for k in ("x", "type", "neigh_ptr", "neigh_idx"):
        dp[k] = torch.from_numpy(np.ascontiguousarray(parts[k])).to(dev)
    colmap = torch.from_numpy(plan.colmap).to(dev)
    rho = torch.from_numpy(parts["rho"]).to(dev)
    vstar = torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)
    own = torch.from_numpy(parts["owner_index"].astype(np.int64)).to(dev)
    vf = hip.compute_volumes(ctx, dp, colmap)
    A, b = hip.assemble_poisson(ctx, dp, colmap, spec.dt, rho, vstar, vfrac=vf[own].contiguous(), ncol=plan.ncol)
    x = torch.randn(plan.ncol, dtype=torch.float64, device=dev)
    y = torch.empty(parts["nlocal"], dtype=torch.float64, device=dev)
    info = A.info()
    alg = 12 * info["nnz"] + 16 * info["nrow"] + 4 * (info["nrow"] + 1)
    for v in (0, 1, 2, 3, 4, 5, 0):
        A.spmv_time(x, y, reps=10, variant=v)
        ms = min(A.spmv_time(x, y, reps=50, variant=v) for _ in range(3))
        print("variant %d: %.4f ms  %.0f GB/s  %.3f of 8 TB/s" % (v, ms, alg / ms / 1e6, alg / ms / 1e6 / 8000), flush=True)
else:
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], check=True)
