#!/bin/bash
# PMC counters of one kernel while a script runs (tuning aid).  usage on the GPU box:
#   bash scripts/pmc_script.sh "<counters>" tag kernel-regex scripts/some_script.py
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
C=$1; T=$2; K=$3; S=$4
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_s_$T -- python3 $GRAFT_REPO_ROOT/$S > $OUT/pmc_s_$T.log 2>&1 || { tail -5 $OUT/pmc_s_$T.log; exit 1; }
F=$(ls -t $OUT/pmc_s_$T/*/*counter_collection.csv | head -1)
python3 - "$F" "$K" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Kernel_Name"]):
        a = acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (kn, c), (n, v) in sorted(acc.items()):
    print("%-40s %-20s launches=%4d avg=%16.1f" % (kn, c, n, v / n))
PY
