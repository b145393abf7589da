import sys, threading
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/oracle']
import numpy as np, scipy.sparse as sps
import isph_amd
from isph_amd import dist, hip, workload
import oracle as orc
import test_gpu_ranks as T
from ranks import RankGroup
lock = threading.Lock()
serial = len(sys.argv) > 1 and sys.argv[1] == "serial"
def body(rank, G, dim, pgrid, n, fill, combine):
    st = T._rank_setup(rank, G, dim, pgrid, n, orc.NULLSPACE)
    ctx, A, plan = st["ctx"], st["A"], st["plan"]
    rp, ci, v = st["csr"]
    rpe, cie, ve = dist.extend_rows(plan, rp, ci, v, G.td(rank))
    Aext = hip.Matrix.from_csr(ctx, rpe, cie, ve)
    if serial: lock.acquire()
    M = hip.PrecondOverlap(ctx, Aext, plan, level_of_fill=fill, combine=combine)
    if serial: lock.release()
    r = np.cos(0.37 * st["rtag"].astype(np.float64))
    # inner schwarz alone on extended vector: use a PrecondSchwarz on Aext
    rext_probe = np.cos(0.11*np.arange(len(rpe)-1))
    if "early" in sys.argv:
        Aext.close()
        z = M.apply(r)
        zs = rext_probe
    else:
        Ms = hip.PrecondSchwarz(ctx, Aext, level_of_fill=fill, overlap=0, block_size=0)
        zs = Ms.apply(rext_probe)
        z = M.apply(r)
        Ms.close(); Aext.close()
    z2 = M.apply(r)
    print("rank", rank, "second apply differs by", np.max(np.abs(z2 - z)))
    M.close(); A.close(); ctx.close()
    return dict(st, ext=(rpe,cie,ve), r=r, z=z, zs=zs, probe=rext_probe, ctx=None, A=None, parts=None)
G = RankGroup(2)
res = G.run(body, 3, (2,1,1), 8, 0, "add")
G.close()
O = T.GlobalOracle(3, (2,1,1), 8, orc.NULLSPACE, [r["rtag"] for r in res])
rglob = np.concatenate([q["r"] for q in res])
zsum = np.zeros(O.N)
for rank, q in enumerate(res):
    rpe,cie,ve = q["ext"]; nl=q["nl"]; plan=q["plan"]
    next_ = len(rpe)-1
    # global (concatenated) index of each extended row
    gidx = np.empty(next_, dtype=np.int64)
    gidx[:nl] = O.off[rank] + np.arange(nl)
    for k,p in enumerate(plan.peers):
        r0,r1 = plan.recv_ptr[k], plan.recv_ptr[k+1]
        gidx[nl+r0:nl+r1] = O.off[int(p)] + plan.recv_idx[r0:r1]
    Ae = sps.csr_matrix((ve,cie,rpe),shape=(next_,next_))
    sub = O.Ap[gidx][:,gidx]
    print("rank",rank,"Aext vs global restriction:", abs(Ae-sub).max(), "ghost sorted:", np.all(np.diff(gidx[nl:])>0))
    F = orc.ILU(rpe,cie,ve,0)
    print("  inner schwarz apply vs ILU oracle:", np.max(np.abs(q["zs"]-F.apply(q["probe"])))/np.abs(q["zs"]).max())
    zext = F.apply(rglob[gidx])
    zsum[gidx] += zext
z = np.concatenate([q["z"] for q in res])
print("device vs composition:", np.max(np.abs(z-zsum))/np.abs(zsum).max())
S = orc.Schwarz(O.Ap.indptr, O.Ap.indices, O.Ap.data, 0, own_ptr=O.off.astype(np.int32), overlap=1, combine="add")
zo = S.apply(rglob)
print("oracle Schwarz vs composition:", np.max(np.abs(zo-zsum))/np.abs(zsum).max(), "vs device", np.max(np.abs(zo-z))/np.abs(zsum).max())
rows, lp, frp, fci, fv = S.export()
print("oracle loc_ptr", lp, "off", O.off)
for rank, q in enumerate(res):
    nl=q["nl"]; plan=q["plan"]
    rpe=q["ext"][0]; next_=len(rpe)-1
    gidx = np.empty(next_, dtype=np.int64)
    gidx[:nl] = O.off[rank] + np.arange(nl)
    for k,p in enumerate(plan.peers):
        r0,r1 = plan.recv_ptr[k], plan.recv_ptr[k+1]
        gidx[nl+r0:nl+r1] = O.off[int(p)] + plan.recv_idx[r0:r1]
    orows = rows[lp[rank]:lp[rank+1]]
    print("rank", rank, "extended rows equal:", len(orows)==len(gidx) and np.array_equal(orows, gidx), len(orows), len(gidx))
