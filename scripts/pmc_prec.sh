#!/bin/bash
# PMC counters of the ILU apply kernel (tuning aid).  usage on the GPU box: bash scripts/pmc_prec.sh "<counters>" tag
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT/pmc_prec_$2 -- python3 $GRAFT_REPO_ROOT/scripts/time_prec.py > $OUT/pmc_prec_$2.log 2>&1 || { tail -5 $OUT/pmc_prec_$2.log; exit 1; }
F=$(ls -t $OUT/pmc_prec_$2/*/*counter_collection.csv | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_ilu_solve_stream" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print("%-28s launches=%4d avg=%16.1f" % (k, n, v / n))
PY
