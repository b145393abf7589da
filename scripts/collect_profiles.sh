#!/bin/bash
# rocprofv3 evidence for the round: kernel stats of the default bench + PMC traffic of the SpMV kernel.
# usage (on the GPU box): bash scripts/collect_profiles.sh   -> writes under gpurun_out/
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-alt --no-orders --no-step > $OUT/prof_stats.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-alt --no-orders --no-step > $OUT/pmc_$C.log 2>&1 || exit 1
done
echo done
