#!/bin/bash
# rocprofv3 kernel stats of the time-step workload.  usage on the GPU box: bash scripts/prof_step.sh <tag> <bench.py step arguments...>
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --workload step --steps 3 --warmup 1 "$@" > $OUT/prof_step_$TAG.log 2>&1 || exit 1
F=$(ls -t $OUT/prof_step_$TAG/*/*kernel_stats.csv | head -1)
python3 - "$F" <<'PY' > $OUT/prof_step_${TAG}_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
steps = 4.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("all kernels: %.1f ms per step" % tot)
for r in rows[:45]:
    print("%-58s calls/step=%7.1f avg_us=%9.1f ms/step=%8.3f" % (r["Name"][:58], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps))
PY
cat $OUT/prof_step_${TAG}_summary.txt
