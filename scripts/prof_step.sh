#!/bin/bash
# rocprofv3 kernel stats of scripts/time_step.py (whole pressure-correction steps).  usage on the GPU box: bash scripts/prof_step.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step -- python3 $GRAFT_REPO_ROOT/scripts/time_step.py > $OUT/prof_step.log 2>&1 || exit 1
F=$(ls -t $OUT/prof_step/*/*kernel_stats.csv | head -1)
python3 - "$F" <<'PY' > $OUT/prof_step_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
steps = 4.0
for r in rows[:60]:
    print("%-56s calls/step=%7.1f avg_us=%9.1f ms/step=%7.3f" % (r["Name"][:56], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps))
PY
echo done
