"""Offline model (CPU only): steps / chunks / cross-DPP-row carries of the triangular-solve stream for the current greedy
packing and for a packing that keeps every row's lanes inside one 16-lane DPP row."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import isph_amd
from isph_amd import workload
from problems import Problem, tgv_spec
from ilu_pack_model import levels


def t_for(grp):
    T = max(1, (sum(grp) + 63) // 64)
    while sum((d + T - 1) // T for d in grp) > 64: T += 1
    return T


def cur_steps(ds, hi=0.9, lo=0.5):
    """k_ilu_schedule: runs of 64 ranks, greedy growth (hi .9, lo .5). returns list of (T, rows, need)"""
    ds = sorted(ds, reverse=True); out = []
    for r0 in range(0, len(ds), 64):
        run = ds[r0:r0 + 64]; a = 0; k = len(run)
        while a < k:
            e = a + 1; S = run[a]; T = (S + 63) // 64; lanes = (S + T - 1) // T
            while e < k:
                de = run[e]; T2 = T; lanes2 = lanes + (de + T - 1) // T
                if lanes2 > 64:
                    while True:
                        T2 += 1
                        lanes2 = sum((d + T2 - 1) // T2 for d in run[a:e + 1])
                        if lanes2 <= 64: break
                    if not (S < hi * 64 * T and (S + de) >= lo * 64 * T2): break
                T = T2; lanes = lanes2; S += de; e += 1
            pos = 0; need = 0
            for d in run[a:e]:
                g = (d + T - 1) // T
                for r in range((pos >> 4) + 1, ((pos + g - 1) >> 4) + 1): need |= 1 << (r - 1)
                pos += g
            out.append((T, e - a, bin(need).count("1")))
            a = e
    return out


def aligned_steps(ds, grow=True):
    """every row inside one DPP row (16 lanes): T >= ceil(d/16); first-fit into 4 bins"""
    ds = sorted(ds, reverse=True); out = []; a = 0; n = len(ds)
    while a < n:
        T = (ds[a] + 15) // 16
        def fit(rows, T):
            bins = [16, 16, 16, 16]
            for d in rows:
                g = (d + T - 1) // T
                for b in range(4):
                    if bins[b] >= g: bins[b] -= g; break
                else: return False
            return True
        e = a + 1
        while e < n:
            if fit(ds[a:e + 1], T): e += 1; continue
            if grow:
                S = sum(ds[a:e]); T2 = T + 1
                while not fit(ds[a:e + 1], T2): T2 += 1
                if 10 * S < 9 * 64 * T and 2 * (S + ds[e]) >= 64 * T2: T = T2; e += 1; continue
            break
        out.append((T, e - a, 0))
        a = e
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    pr = Problem(tgv_spec(dim=3, n=n, mode=workload.ADVECT, brick=8))
    rp, ci = pr.P.graph()
    B = 512
    variants = [("cur", cur_steps), ("aligned", aligned_steps)]
    for hi in (0.95, 1.0, 1.01, 2.0):
        for lo in (0.5, 0.6, 0.65, 0.7):
            variants.append(("g_%.2f_%.2f" % (hi, lo), (lambda h, l: (lambda d: cur_steps(d, h, l)))(hi, lo)))
    res = {k: [0, 0, 0, 0] for k, _ in variants}
    nnz_off = 0; nblk = 0; nlev = 0
    for b in range(0, min(pr.n, 16 * B), B):
        nblk += 1
        for direction in (0, 1):
            deps = []
            for i in range(b, min(b + B, pr.n)):
                c = ci[rp[i]:rp[i + 1]]
                c = c[(c >= b) & (c < b + B)]
                deps.append((c[c < i] - b) if direction == 0 else (c[c > i] - b))
            if direction == 1:
                m = len(deps)
                deps = [np.array([m - 1 - j for j in d]) for d in deps[::-1]]
            lev = levels(deps)
            nlev += lev.max()
            for l in range(1, lev.max() + 1):
                ds = [len(deps[i]) for i in np.nonzero(lev == l)[0]]
                nnz_off += sum(ds)
                for name, fn in variants:
                    st = fn(ds)
                    r = res[name]
                    r[0] += sum(s[0] for s in st); r[1] += len(st); r[2] += sum(s[2] for s in st); r[3] += sum(1 for s in st if s[2])
    print("blocks", nblk, "entries/block", nnz_off / nblk, "levels/block (L+U)", nlev / nblk)
    for k, r in res.items():
        print("%-15s chunks/block %7.1f  padding x%.3f  steps/block %6.1f  carry stages/block %6.1f  steps with carries %6.1f"
              % (k, r[0] / nblk, r[0] * 64 / nnz_off, r[1] / nblk, r[2] / nblk, r[3] / nblk))
