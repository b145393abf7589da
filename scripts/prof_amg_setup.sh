#!/bin/bash
# rocprofv3 kernel trace of scripts/time_amg.py: the kernels of ONE hierarchy set-up at 100^3, in launch order with their
# durations (the last create of the run).  usage on the GPU box: bash scripts/prof_amg_setup.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_amg_setup -- python3 $GRAFT_REPO_ROOT/scripts/time_amg.py > $OUT/prof_amg_setup.log 2>&1 || exit 1
F=$(ls $OUT/prof_amg_setup/*/*kernel_trace.csv | head -1)
python3 - "$F" <<'PY' > $OUT/prof_amg_setup_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the set-ups start at k_sell_to_csr_i32; the last one ends before the first Krylov kernel that follows it
starts = [i for i, r in enumerate(rows) if "k_sell_to_csr_i32" in r["Kernel_Name"]]
i0 = starts[-1]
i1 = next(i for i in range(i0, len(rows)) if "k_multi_" in rows[i]["Kernel_Name"] or "k_nrm" in rows[i]["Kernel_Name"] or "k_ilu_solve_stream" in rows[i]["Kernel_Name"])
agg = collections.OrderedDict()
for r in rows[i0:i1]:
    n = r["Kernel_Name"].split("(")[0][-60:]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(n, [0, 0.0, 0.0])
    a[0] += 1; a[1] += d; a[2] = max(a[2], d)
tot = sum(a[1] for a in agg.values())
span = (int(rows[i1 - 1]["End_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e3
print("one SA-AMG set-up at 100^3: %d launches, %.2f ms inside kernels, %.2f ms first start to last end" % (i1 - i0, tot / 1e3, span / 1e3))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s calls=%3d total_us=%8.1f max_us=%8.1f" % (n, a[0], a[1], a[2]))
PY
cat $OUT/prof_amg_setup_summary.txt
