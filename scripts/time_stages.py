import sys, time, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import isph_amd
from isph_amd import hip, workload, dist
dev=torch.device("cuda",0)
ctx=hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
spec=workload.TGVSpec(dim=3, ncell=(100,100,100), brick=(8,8,8), mode=workload.ADVECT)
t=time.perf_counter(); parts=workload.make_tgv(spec); print("generate %.1f ms"%((time.perf_counter()-t)*1e3))
plan=dist.make_plan(parts,None)
dp=dict(parts)
for k in ("x","type","neigh_ptr","neigh_idx"): dp[k]=torch.from_numpy(np.ascontiguousarray(parts[k])).to(dev)
colmap=torch.from_numpy(plan.colmap).to(dev); rho=torch.from_numpy(parts["rho"]).to(dev); vstar=torch.from_numpy(np.ascontiguousarray(parts["v"])).to(dev)
own=torch.from_numpy(parts["owner_index"].astype(np.int64)).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    vf=hip.compute_volumes(ctx,dp,colmap); torch.cuda.synchronize(); t1=time.perf_counter()
    vfrac=vf[own].contiguous(); torch.cuda.synchronize(); t2=time.perf_counter()
    A,b=hip.assemble_poisson(ctx,dp,colmap,spec.dt,rho,vstar,vfrac=vfrac,ncol=plan.ncol); torch.cuda.synchronize(); t3=time.perf_counter()
    M=hip.Precond(ctx,A,"bjacobi-ilu0",int(os.environ.get("ISPH_BLOCK","512"))); ctx.sync(); t4=time.perf_counter()
    M.close(); t5=time.perf_counter(); A.close(); t6=time.perf_counter()
    print("rep %d: volumes %.1f  fwd %.1f  assemble %.1f  ilu_create %.1f  prec_destroy %.1f mat_destroy %.1f ms"%(rep,(t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t5-t4)*1e3,(t6-t5)*1e3))
