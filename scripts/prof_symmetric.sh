#!/bin/bash
# kernel statistics of the Symmetric-family computePre + assembly (scripts/time_symmetric.py).  usage on the GPU box: bash scripts/prof_symmetric.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_sym
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sym -- python3 $GRAFT_REPO_ROOT/scripts/time_symmetric.py > $OUT/prof_sym.log 2>&1 || { tail -5 $OUT/prof_sym.log; exit 1; }
grep "^rep" $OUT/prof_sym.log
python3 $GRAFT_REPO_ROOT/scripts/kstats.py $(ls $OUT/prof_sym/*/*kernel_stats.csv | head -1) "correction|asm_poisson|volumes|asm_count"
