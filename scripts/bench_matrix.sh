#!/bin/bash
# the size / preconditioner table of DESIGN.md section 7 (one line per run).  usage on the GPU box: bash scripts/bench_matrix.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
: > $OUT/bench_matrix.txt
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin "$@" > $OUT/bench_matrix_one.log 2>&1
  grep '^{"metric' $OUT/bench_matrix_one.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$*', '| rows', c['rows_per_gpu'], 'its', c['iterations'], 'ms', round(d['ms_per_step'],1), 'spmv frac', round(d['roofline']['frac'],3), 'levels', [l['rows'] for l in c.get('amg',{}).get('levels',[])])" >> $OUT/bench_matrix.txt 2>&1 || echo "$* FAILED" >> $OUT/bench_matrix.txt
}
run --prec bjacobi-ilu0
run --prec sa-amg
run --prec sa-amg --amg-theta 0.02
run --prec jacobi
run --prec bjacobi-ilu1
run --prec bjacobi-ilu2
run --prec bjacobi-ilu0 --block 256
run --ncell 126 --prec bjacobi-ilu0
run --ncell 126 --prec sa-amg
run --ncell 160 --prec bjacobi-ilu0
run --ncell 160 --prec sa-amg
run --ncell 64 --kernel quintic --prec bjacobi-ilu0
run --ncell 64 --kernel quintic --prec sa-amg
run --ncell 100 --kernel quintic --prec bjacobi-ilu0
run --ncell 100 --kernel quintic --prec sa-amg
run --mode jitter --prec bjacobi-ilu0
run --mode jitter --prec jacobi
run --mode jitter --prec sa-amg
run --prec ilu0 --steps 1 --warmup 0
cat $OUT/bench_matrix.txt
