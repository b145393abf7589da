"""Offline model of the ILU triangular-solve stream packing (tuning aid, CPU only).
Builds the bench matrix pattern at a small size, levels each 512-row block and counts 64-lane chunks
for packing strategies."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import isph_amd
from isph_amd import workload
from problems import Problem, tgv_spec

def pow2ceil(v):
    p = 1
    while p < v: p *= 2
    return p

def levels(deps):
    lev = np.zeros(len(deps), int)
    for i, d in enumerate(deps):          # lower: deps < i ; processed in order
        lev[i] = 1 + max((lev[j] for j in d), default=-1) if len(d) else 0
    return lev

def pack_current(ds):
    ds = sorted(ds, reverse=True); tot = 0
    for s in range(0, len(ds), 8):
        grp = ds[s:s + 8]; r = len(grp)
        G = 8 if r > 4 else 16 if r > 2 else 32 if r > 1 else 64
        tot += max((d + G - 1) // G for d in grp)
    return tot

def pack_var(ds, gmin=8):
    """rows sorted desc; greedy steps; in each step choose minimal T with sum pow2ceil(ceil(d/T)) <= 64"""
    ds = sorted(ds, reverse=True); tot = 0; i = 0; n = len(ds)
    best_total = None
    # dynamic programming over split points (levels are small)
    INF = 10 ** 9
    cost = [INF] * (n + 1); cost[0] = 0
    for a in range(n):
        if cost[a] == INF: continue
        for b in range(a + 1, min(n, a + 64 // gmin) + 1):
            grp = ds[a:b]
            T = max(1, (sum(grp) + 63) // 64)
            while True:
                lanes = sum(max(gmin, pow2ceil((d + T - 1) // T)) for d in grp)
                if lanes <= 64 and max(max(gmin, pow2ceil((d + T - 1) // T)) for d in grp) <= 64: break
                T += 1
            cost[b] = min(cost[b], cost[a] + T)
    return cost[n]

def pack_any(ds):
    """arbitrary contiguous lane counts g_i >= 1 per row, one step per <=64 rows: T = min T with sum ceil(d/T) <= 64"""
    ds = sorted(ds, reverse=True); tot = 0
    for s0 in range(0, len(ds), 64):
        grp = ds[s0:s0 + 64]
        T = max(1, (sum(grp) + 63) // 64)
        while sum((d + T - 1) // T for d in grp) > 64: T += 1
        tot += T
    return tot

def t_for(grp):
    T = max(1, (sum(grp) + 63) // 64)
    while sum((d + T - 1) // T for d in grp) > 64: T += 1
    return T

def pack_greedy2(ds, hi=0.9, lo=0.6):
    ds = sorted(ds, reverse=True); tot = 0; a = 0; n = len(ds)
    while a < n:
        b = a + 1; T = t_for(ds[a:b])
        while b < n and b - a < 64:
            T2 = t_for(ds[a:b + 1])
            if T2 != T and not (sum(ds[a:b]) < hi * 64 * T and sum(ds[a:b + 1]) >= lo * 64 * T2): break
            T = T2; b += 1
        tot += T; a = b
    return tot

def pack_greedy(ds, theta=0.75):
    """rows sorted desc; a step grows while its chunk count stays, or while it is still poorly filled"""
    ds = sorted(ds, reverse=True); tot = 0; a = 0; n = len(ds)
    while a < n:
        b = a + 1; T = t_for(ds[a:b])
        while b < n and b - a < 64:
            T2 = t_for(ds[a:b + 1])
            if T2 != T and sum(ds[a:b]) >= theta * 64 * T: break
            T = T2; b += 1
        tot += T; a = b
    return tot

def pack_dp(ds, kmax=64):
    ds = sorted(ds, reverse=True); n = len(ds)
    INF = 10 ** 9
    cost = [INF] * (n + 1); cost[0] = 0
    for a in range(n):
        for b in range(a + 1, min(n, a + kmax) + 1):
            cost[b] = min(cost[b], cost[a] + t_for(ds[a:b]))
    return cost[n]

def pack_hybrid(ds):
    return pack_dp(ds) if len(ds) <= 12 else pack_greedy(ds, 0.75)

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    pr = Problem(tgv_spec(dim=3, n=n, mode=workload.ADVECT, brick=8))
    rp, ci = pr.P.graph()
    B = 512
    tot = {"ideal": 0, "cur": 0, "var8": 0, "var4": 0, "any": 0, "greedy": 0, "greedy90": 0, "dp": 0, "hybrid": 0, "greedy2": 0}
    nnz_off = 0
    hist = {}
    for b in range(0, min(pr.n, 16 * B), B):
        for direction in (0, 1):
            rows = range(b, min(b + B, pr.n))
            deps = []
            for i in rows:
                c = ci[rp[i]:rp[i + 1]]
                c = c[(c >= b) & (c < b + B)]
                d = (c[c < i] - b) if direction == 0 else (c[c > i] - b)
                deps.append(d)
            if direction == 1:     # upper: reverse order
                m = len(deps)
                deps = [np.array([m - 1 - j for j in d]) for d in deps[::-1]]
            lev = levels(deps)
            for l in range(1, lev.max() + 1):
                ds = [len(deps[i]) for i in np.nonzero(lev == l)[0]]
                hist[len(ds)] = hist.get(len(ds), 0) + 1
                nnz_off += sum(ds)
                tot["ideal"] += (sum(ds) + 63) // 64
                tot["cur"] += pack_current(ds)
                tot["var8"] += pack_var(ds, 8)
                tot["var4"] += pack_var(ds, 4)
                tot["any"] += pack_any(ds)
                tot["greedy"] += pack_greedy(ds)
                tot["greedy90"] += pack_greedy(ds, 0.9)
                tot["dp"] += pack_dp(ds)
                tot["hybrid"] += pack_hybrid(ds)
                tot["greedy2"] += pack_greedy2(ds)
                for hi in (0.9, 0.95, 1.0):
                    for lo in (0.3, 0.4, 0.5):
                        tot.setdefault("g2_%.2f_%.1f" % (hi, lo), 0)
                        tot["g2_%.2f_%.1f" % (hi, lo)] += pack_greedy2(ds, hi, lo)
    print("off-diag entries", nnz_off, "-> min chunks", nnz_off / 64)
    for k, v in tot.items(): print(k, v, "padding x%.3f" % (v * 64 / nnz_off))
    print("rows/level histogram", sorted(hist.items()))

    for name, ds in (("63x1+100", [1] * 63 + [100]), ("200x2", [2] * 200), ("20x40", [40] * 20), ("3,3,60", [3, 3, 60])):
        print(name, "ideal", (sum(ds) + 63) // 64, "any", pack_any(ds), "greedy", pack_greedy(ds), "dp", pack_dp(ds), "hybrid", pack_hybrid(ds), "greedy2", pack_greedy2(ds), "cur", pack_current(ds))
    # scan greedy2 thresholds on the real pattern
