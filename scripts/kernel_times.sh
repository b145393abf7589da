#!/bin/bash
# average kernel times of the default bench matching a pattern.  usage on the GPU box: bash scripts/kernel_times.sh <pattern> [bench args]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
PAT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/prof_kt.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/scripts/kstats.py $(ls $OUT/prof_kt/*/*kernel_stats.csv | head -1) "$PAT"
grep '^{"metric' $OUT/prof_kt.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'its', d['config']['iterations'])"
