#!/bin/bash
# rocprofv3 kernel trace of scripts/prof_assembly.py: the kernels of the LAST assemble_poisson call in launch order with
# their durations and the idle gap before each.  usage on the GPU box: bash scripts/prof_assembly_timeline.sh [bricks|caller]
set -o pipefail
MODE=${1:-bricks}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_asmtl_$MODE -- python3 $GRAFT_REPO_ROOT/scripts/prof_assembly.py $MODE > $OUT/prof_asmtl_$MODE.log 2>&1 || exit 1
F=$(ls -t $OUT/prof_asmtl_$MODE/*/*kernel_trace.csv | head -1)
python3 - "$F" <<'PY' > $OUT/prof_asmtl_${MODE}_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a call ends with k_sum_rowlen / the right-hand side scatter; it starts after the previous call's last kernel: split on
# the long host gaps (> 300 us: torch.cuda.synchronize + the timer) and take the last group with an assembly row kernel
groups, cur, prev_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s - prev_end > 300000 and cur:
        groups.append(cur); cur = []
    cur.append(r); prev_end = e
if cur: groups.append(cur)
g = [x for x in groups if any("k_asm_poisson" in r["Kernel_Name"] for r in x)][-1]
t0 = int(g[0]["Start_Timestamp"]); prev = None; tot = 0.0
for r in g:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev is None else (s - prev) / 1e3
    d = (e - s) / 1e3; tot += d
    print("%9.1f us  gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, gap, d, r["Kernel_Name"].split("(")[0][-70:]))
    prev = e
print("kernels %.3f ms, first start to last end %.3f ms, %d launches" % (tot / 1e3, (prev - t0) / 1e6, len(g)))
PY
grep assemble_poisson $OUT/prof_asmtl_$MODE.log >> $OUT/prof_asmtl_${MODE}_summary.txt
cat $OUT/prof_asmtl_${MODE}_summary.txt
