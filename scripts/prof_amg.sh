#!/bin/bash
# rocprofv3 kernel stats of the SA-AMG bench (set-up + cycle kernels).  usage on the GPU box: bash scripts/prof_amg.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_amg -- python3 $GRAFT_REPO_ROOT/bench.py --prec sa-amg --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-alt --no-orders --no-step > $OUT/prof_amg.log 2>&1 || exit 1
F=$(ls $OUT/prof_amg/*/*kernel_stats.csv | head -1)
python3 - "$F" <<'PY' > $OUT/prof_amg_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
steps = 4.0
for r in rows[:45]:
    print("%-52s calls/solve=%7.1f avg_us=%9.1f ms/solve=%7.3f" % (r["Name"][:52], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps))
PY
echo done
