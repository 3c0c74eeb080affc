#!/bin/bash
# Diagnostic sweep of the tiled solve's placement knobs (CAVIAR_HIP_XCD_MODE, CAVIAR_HIP_SOLVE_GROUP) on the default bench workload.
# usage: tools/solve_knobs.sh OUTDIR
out=${1:-gpurun_out/knobs}
mkdir -p "$out"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-single-leg --streams-leg 0 --no-cpu-baseline --steps 6 --warmup 2 > "$out/$name.json" 2> "$out/$name.err" || return 1
  python tools/bench_summary.py "$out/$name.json"
}
for m in ${2:-1 2 3}; do run m$m CAVIAR_HIP_XCD_MODE=$m || exit 1; done
