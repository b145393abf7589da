import numpy as np, sys
b = open(sys.argv[1], "rb").read()
n4, nrun = np.frombuffer(b[:16], np.int64)
o = 16
t = np.frombuffer(b[o:o + 8 * n4], np.int64).astype(np.float64) * 0.01; o += 8 * n4
ts = np.frombuffer(b[o:o + 8 * n4], np.int64).astype(np.float64) * 0.01; o += 8 * n4
t2 = np.frombuffer(b[o:o + 8 * n4], np.int64).astype(np.float64) * 0.01; o += 8 * n4
lv = np.frombuffer(b[o:o + 8 * n4], np.int64); o += 8 * n4
rs = np.frombuffer(b[o:o + 4 * (nrun + 1)], np.int32)
ok = lv >= 0
nl = lv.max() + 1
done = np.zeros(nl); np.maximum.at(done, lv[ok], t[ok])
seen = np.zeros(nl); np.maximum.at(seen, lv[ok], ts[ok])
ph2 = np.zeros(nl); np.maximum.at(ph2, lv[ok], t2[ok])
pos = np.arange(n4); run_of = np.searchsorted(rs, pos, side="right") - 1
lvl_run = np.zeros(nl, np.int64); np.maximum.at(lvl_run, lv[ok], run_of[ok])
idx = np.zeros(nl, np.int64)
for l in range(1, nl): idx[l] = idx[l - 1] + 1 if lvl_run[l] == lvl_run[l - 1] else 0
hop = np.diff(done)
inr = idx[1:] > 0
print("levels", nl, "runs", nrun, "mean hop %.3f; in-run median %.3f mean %.3f; cross median %.3f" % (hop.mean(), np.median(hop[inr]), hop[inr].mean(), np.median(hop[~inr])))
print("in-run: prev level done -> this level enters phase 2: median %.3f (negative: was waiting)" % np.median((ph2[1:] - done[:-1])[inr]))
print("in-run: max(enter, prev done) -> seen: median %.3f ; seen -> done: median %.3f" % (np.median((seen[1:] - np.maximum(ph2[1:], done[:-1]))[inr]), np.median((done - seen)[1:][inr])))
late = (ph2[1:] - done[:-1])[inr]
print("share of in-run hops where the level entered phase 2 AFTER the previous level was done: %.2f; their lateness median %.2f" % ((late > 0).mean(), np.median(late[late > 0]) if (late > 0).any() else 0))
