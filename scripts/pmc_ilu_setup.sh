#!/bin/bash
# PMC counters of the three block-ILU(0) set-up kernels on the bench matrix (scripts/time_ilu_setup.py), one rocprofv3 pass
# per counter group (no trace domains beside --kernel-trace).  usage on the GPU box: bash scripts/pmc_ilu_setup.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SUM=$OUT/r05_pmc_ilu_setup.txt
: > $SUM
G=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE MemUnitStalled"; do
  G=$((G+1))
  ISPH_REPS=2 timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_ilu_$G -- python3 $GRAFT_REPO_ROOT/scripts/time_ilu_setup.py > $OUT/pmc_ilu_$G.log 2>&1 || { echo "group $G ($C) failed" >> $SUM; tail -3 $OUT/pmc_ilu_$G.log >> $SUM; continue; }
  F=$(ls -t $OUT/pmc_ilu_$G/*/*counter_collection.csv | head -1)
  python3 - "$F" >> $SUM <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"]
    for k in ("k_ilu_extract", "k_ilu_schedule", "k_ilu_factor", "k_ilu_solve_stream"):
        if k in kn:
            a = acc[(k, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (kn, c), (n, v) in sorted(acc.items()):
    print("%-20s %-24s launches=%3d avg=%18.1f" % (kn, c, n, v / n))
PY
done
cat $SUM
