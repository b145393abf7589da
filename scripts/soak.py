"""Soak: many consecutive set-up + solve rounds of the headline configuration (ILU, SA-AMG and the host-CSR drop-in in turn); device memory
in use and the time per round must stay flat (the pool hands the same buffers back every round).
usage on the GPU box: python scripts/soak.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import isph_amd  # noqa: F401
from isph_amd import hip, workload, dist

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = hip.Context(0, stream=st.cuda_stream)
spec = workload.TGVSpec(dim=3, ncell=(100, 100, 100), brick=(8, 8, 8), mode=workload.ADVECT)
parts = workload.make_tgv(spec)
plan = dist.make_plan(parts, None)
n = parts["nlocal"]
dp = dict(parts)
for k in ("x", "type", "neigh_ptr", "neigh_idx"):
    dp[k] = torch.from_numpy(np.ascontiguousarray(parts[k])).to(dev)
colmap = torch.from_numpy(plan.colmap).to(dev)
own = torch.from_numpy(parts["owner_index"].astype(np.int64)).to(dev)
rho = torch.from_numpy(parts["rho"]).to(dev)
v = torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)
nullvec = torch.full((n,), 1.0 / np.sqrt(n), dtype=torch.float64, device=dev)
times, mem = [], []
host = None   # the system as a host CSR: every third round goes through the drop-in's ingress (pinned ring, fused set-up)
for r in range(rounds):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if r % 3 == 2 and host is not None:
        A, M = hip.Matrix.from_host_csr_with_bjacobi(ctx, host[0], host[1], host[2], 512)
        bh, xh = host[3].copy(), np.zeros(n)
        info = hip.solve(ctx, A, bh, xh, prec=M, singular=True)
    else:
        vf = hip.compute_volumes(ctx, dp, colmap)
        A, b = hip.assemble_poisson(ctx, dp, colmap, spec.dt, rho, v, vfrac=vf[own].contiguous())
        if host is None:
            host = A.export_csr() + (b.cpu().numpy(),)
        M = hip.PrecondAMG(ctx, A, nullvec=nullvec) if r % 2 else hip.Precond(ctx, A, "bjacobi-ilu0", 512)
        x = torch.zeros(n, dtype=torch.float64, device=dev)
        info = hip.solve(ctx, A, b, x, prec=M, singular=True)
    M.close()
    A.close()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
    fr, tot = torch.cuda.mem_get_info()
    mem.append((tot - fr) / 1e9)
    assert info.converged == 1
    if r % 50 == 49 or r == rounds - 1:
        print("round %d: last 20 rounds %.1f ms avg (ILU, AMG and the host-ingress drop-in in turn), device memory in use %.2f GB, library pool %.2f GB"
              % (r + 1, float(np.mean(times[-20:])), mem[-1], hip.pool_cached_bytes() / 1e9), flush=True)
print("memory in use: round 10 %.2f GB, last %.2f GB; time per round: rounds 10-30 %.1f ms, last 20 %.1f ms"
      % (mem[9], mem[-1], float(np.mean(times[10:30])), float(np.mean(times[-20:]))))
assert mem[-1] <= mem[9] + 0.05, "device memory grows from round to round"
