#!/bin/bash
# rocprofv3 kernel trace of the RCCL halo path on one GPU (periodic images routed through send/recv-to-self):
# evidence that the interior slices of the SpMV run while the exchange is in flight.
# usage (GPU box): bash scripts/prof_overlap.sh  -> gpurun_out/prof_overlap/ + gpurun_out/r02_halo_overlap.txt
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_overlap
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_overlap -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --force-rccl > $OUT/prof_overlap.log 2>&1 || { tail -5 $OUT/prof_overlap.log; exit 1; }
python3 $GRAFT_REPO_ROOT/scripts/halo_overlap.py $(ls $OUT/prof_overlap/*/*kernel_trace.csv | head -1) > $OUT/r02_halo_overlap.txt
cat $OUT/r02_halo_overlap.txt
