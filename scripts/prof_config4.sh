#!/bin/bash
# kernel statistics of the configs[4]-shaped run (bcc/Quintic, SA-AMG).  usage on the GPU box: bash scripts/prof_config4.sh <cells> [prec]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_c4
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -- python3 $GRAFT_REPO_ROOT/scripts/run_config4.py "$@" > $OUT/prof_c4.log 2>&1 || { tail -5 $OUT/prof_c4.log; exit 1; }
grep "^\[\|resid" $OUT/prof_c4.log
python3 - $(ls $OUT/prof_c4/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:28]:
    print("%-60s calls=%5s total_ms=%9.2f avg_us=%10.1f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
