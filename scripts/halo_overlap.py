"""Reads a rocprofv3 kernel trace of `bench.py --force-rccl` and reports, per SpMV with a halo, how the interior-slice
kernel (LIST, no ghost columns) overlaps the RCCL send/recv kernel of the same exchange and how long the boundary kernel
waits.  usage: python scripts/halo_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
is_int = lambda n: "k_sell_spmv16<8, false, true, false>" in n or "k_sell_spmv<8, false, true, true, false>" in n
is_bnd = lambda n: "k_sell_spmv16<8, false, true, true>" in n or "k_sell_spmv<8, false, true, true, true>" in n
is_rccl = lambda n: "rcclGenericKernel" in n or "ncclDevKernel" in n or "SendRecv" in n
ints = [r for r in rows if is_int(r[2])]
bnds = [r for r in rows if is_bnd(r[2])]
rccl = [r for r in rows if is_rccl(r[2])]
print("kernels in trace: %d, interior SpMV launches: %d, boundary: %d, RCCL send/recv kernels: %d" % (len(rows), len(ints), len(bnds), len(rccl)))
if not ints or not rccl:
    names = sorted({r[2][:90] for r in rows if "ccl" in r[2].lower()})
    print("rccl-like kernel names:", names)
    sys.exit(0)
ov_tot, n_ov, int_tot, rc_tot = 0, 0, 0, 0
ri = 0
for s, e, name, q in ints:
    # the exchange kernel that starts closest before/around this interior launch
    best = None
    for rs, re_, rn, rq in rccl:
        if re_ < s - 2_000_000 or rs > e + 2_000_000:
            continue
        ov = max(0, min(e, re_) - max(s, rs))
        if best is None or ov > best[0]:
            best = (ov, rs, re_, rq)
    int_tot += e - s
    if best:
        ov_tot += best[0]
        rc_tot += best[2] - best[1]
        n_ov += best[0] > 0
print("interior launches overlapping an RCCL kernel in time: %d of %d" % (n_ov, len(ints)))
print("mean interior kernel %.1f us, mean RCCL send/recv kernel %.1f us, mean overlap %.1f us" %
      (int_tot / len(ints) / 1e3, rc_tot / max(len(ints), 1) / 1e3, ov_tot / len(ints) / 1e3))
print("queues: interior SpMV on", sorted({r[3] for r in ints}), " RCCL on", sorted({r[3] for r in rccl}))
bw = [r[1] - r[0] for r in bnds]
print("mean boundary kernel %.1f us" % (sum(bw) / max(len(bw), 1) / 1e3))
