import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import isph_amd
from isph_amd import hip, workload
import oracle as orc
from problems import Problem, tgv_spec
ctx = hip.Context(0)
pr = Problem(tgv_spec(dim=3, n=16, mode=workload.ADVECT, brick=8))
rp, ci, val, b = pr.poisson(); n = pr.n
nv = np.ones(n) / np.sqrt(n)
kw = dict(theta=0.0, block=256, coarse_max=64)
G = orc.AMG(rp, ci, val, nullvec=nv, **kw)
A = hip.Matrix.from_csr(ctx, rp, ci, val)
M = hip.PrecondAMG(ctx, A, nullvec=nv, params=hip.AmgParams(**kw))
print([G.level_info(l) for l in range(G.levels)], [M.level_info(l) for l in range(M.levels)])
ao, ag = G.aggregates(0), M.aggregates(0)
print("agg mismatch", (ao != ag).sum(), "nagg", ao.max() + 1, ag.max() + 1, "neg", (ao < 0).sum(), (ag < 0).sum())
ro, co, vo = G.export(0, "P"); rg, cg, vg = M.export(0, "P")
lo, lg = np.diff(ro), np.diff(rg)
bad = np.nonzero(lo != lg)[0]
print("rows with different P length", len(bad), bad[:10])
for i in bad[:3]:
    print(i, "oracle", co[ro[i]:ro[i+1]], "gpu", cg[rg[i]:rg[i+1]], vg[rg[i]:rg[i+1]])
    nb = ci[rp[i]:rp[i+1]]
    print("  neighbour aggs", sorted(set(ao[nb])))

# python emulation of the MIS-2 rounds
def h32(x):
    x = np.uint64(x) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(16); return x
idx = np.arange(n, dtype=np.uint64)
hh = h32(idx)
def mk(state): return (np.uint64(state) << np.uint64(62)) | (hh << np.uint64(30)) | idx
rows = np.repeat(np.arange(n), np.diff(rp))
strong = (ci != rows) & (val != 0)
key = mk(1)
state = np.ones(n, int)
import scipy.sparse as sps
S = sps.csr_matrix((strong.astype(np.int8), ci, rp), shape=(n, n))
S.eliminate_zeros()
def nbmax(k):
    out = k.copy()
    for i in range(n):
        nb = S.indices[S.indptr[i]:S.indptr[i+1]]
        if len(nb): out[i] = max(out[i], k[nb].max())
    return out
rounds = 0
while (state == 1).any():
    key = (state.astype(np.uint64) << np.uint64(62)) | (hh << np.uint64(30)) | idx
    t2 = nbmax(nbmax(key))
    und = state == 1
    newroot = und & (t2 == key)
    cov = und & ~newroot & ((t2 >> np.uint64(62)) == 3)
    state[newroot] = 3; state[cov] = 0
    rounds += 1
print("emulated roots", (state == 3).sum(), "rounds", rounds)
