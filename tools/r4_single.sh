#!/bin/bash
# one circuit at cfg3: per-iteration wall times and the Cholesky families below CM_LARGE_TILES
out=${1:-gpurun_out/r04}; mkdir -p $out
A="--circuits-per-gpu 1 --batch-leg 0 --steps 5 --warmup 2 --no-cpu-baseline"
CAVIAR_HIP_TRACE_ITERS=1 timeout -k 10 200 python bench.py $A --steps 1 --warmup 1 > $out/s_trace.json 2> $out/s_trace.err || exit 1
for cfg in "1 48" "2 8" "2 4" "0 8" "0 4"; do
  set -- $cfg
  CAVIAR_HIP_CHOL_FAMILY=$1 CAVIAR_HIP_LARGE_TILES=$2 timeout -k 10 200 python bench.py $A > $out/s_fam$1_$2.json 2> $out/s_fam$1_$2.err || exit 1
  echo "family $1 large_tiles $2: $(python tools/bench_summary.py $out/s_fam$1_$2.json)"
done
grep "^it " $out/s_trace.err | tail -52 | head -52
