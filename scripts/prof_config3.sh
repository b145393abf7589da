#!/bin/bash
# kernel statistics of the configs[3]-shaped step (scripts/run_config3.py).  usage on the GPU box: bash scripts/prof_config3.sh [cells]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_c3
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 $GRAFT_REPO_ROOT/scripts/run_config3.py "$@" > $OUT/prof_c3.log 2>&1 || { tail -5 $OUT/prof_c3.log; exit 1; }
grep "^step\|^particles" $OUT/prof_c3.log
python3 - $(ls $OUT/prof_c3/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:26]:
    print("%-62s calls=%5s total_ms=%9.2f avg_us=%10.1f" % (r["Name"][:62], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
