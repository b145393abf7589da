#!/bin/bash
# kernel + memory-copy timeline of the SolverLin drop-in (C++ driver, "timed" mode).  usage on the GPU box: bash scripts/prof_dropin.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/scripts/dropin_100.py 100 1 keep > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_dropin
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/prof_dropin -- $GRAFT_REPO_ROOT/tests/cpp/test_solver_lin /dev/shm/isph_dropin_sys.bin /dev/shm/isph_dropin_x.bin 1 timed 2 > $OUT/prof_dropin.log 2>&1
rm -f /dev/shm/isph_dropin_sys.bin /dev/shm/isph_dropin_x.bin
python3 - $OUT/prof_dropin <<'PY'
import csv, glob, sys
d = sys.argv[1]
k = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
m = list(csv.DictReader(open(glob.glob(d + "/*/*memory_copy_trace.csv")[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in k]
big = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Bytes"]) if "Bytes" in r else 0) for r in m]
big = [b for b in big if b[2] > (1 << 20)]
# last ingress = the last run of >1 MiB H2D copies that ends before the last k_ilu_factor burst
last_copy_end = max(b[1] for b in big if b[2] > (20 << 20))
first_copy = [b for b in big if b[2] > (1 << 20) and b[0] > last_copy_end - 40_000_000]
t0 = min(b[0] for b in first_copy)
print("last ingress: first big copy at 0, last big copy ends at %.2f ms" % ((last_copy_end - t0) / 1e6))
for s, e, n in sorted(ev):
    if s > last_copy_end - 1_500_000 and s < last_copy_end + 6_000_000:
        print("  %-46s start %+8.3f ms  dur %7.3f ms" % (n, (s - last_copy_end) / 1e6, (e - s) / 1e6))
PY
