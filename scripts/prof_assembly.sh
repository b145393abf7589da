#!/bin/bash
# rocprofv3 kernel stats of the Poisson assembly (scripts/prof_assembly.py).  usage on the GPU box: bash scripts/prof_assembly.sh [bricks|caller]
set -o pipefail
MODE=${1:-bricks}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_asm_$MODE -- python3 $GRAFT_REPO_ROOT/scripts/prof_assembly.py $MODE > $OUT/prof_asm_$MODE.log 2>&1 || exit 1
F=$(ls -t $OUT/prof_asm_$MODE/*/*kernel_stats.csv | head -1)
python3 - "$F" <<'PY' > $OUT/prof_asm_${MODE}_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
reps = 6.0
tot = 0.0
for r in rows[:40]:
    ms = float(r["TotalDurationNs"]) / 1e6 / reps
    tot += ms
    print("%-60s calls/asm=%6.1f avg_us=%9.1f ms/asm=%7.3f" % (r["Name"][:60], float(r["Calls"]) / reps, float(r["AverageNs"]) / 1e3, ms))
print("sum of the listed kernels per assembly: %.3f ms" % tot)
PY
grep assemble_poisson $OUT/prof_asm_$MODE.log >> $OUT/prof_asm_${MODE}_summary.txt
echo done
