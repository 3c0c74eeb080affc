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
run m1_g0 CAVIAR_HIP_XCD_MODE=1 CAVIAR_HIP_SOLVE_GROUP=0 &&
run m2_g0 CAVIAR_HIP_XCD_MODE=2 CAVIAR_HIP_SOLVE_GROUP=0 &&
run m0_g0 CAVIAR_HIP_XCD_MODE=0 CAVIAR_HIP_SOLVE_GROUP=0 &&
run m1_g32 CAVIAR_HIP_XCD_MODE=1 CAVIAR_HIP_SOLVE_GROUP=32 &&
run m1_g16 CAVIAR_HIP_XCD_MODE=1 CAVIAR_HIP_SOLVE_GROUP=16 &&
run m2_g16 CAVIAR_HIP_XCD_MODE=2 CAVIAR_HIP_SOLVE_GROUP=16 &&
run m1_g8 CAVIAR_HIP_XCD_MODE=1 CAVIAR_HIP_SOLVE_GROUP=8 &&
run m2_g8 CAVIAR_HIP_XCD_MODE=2 CAVIAR_HIP_SOLVE_GROUP=8
