#!/bin/bash
# per-iteration wall times of a 256-circuit lock-step batch (CAVIAR_HIP_TRACE_ITERS)
out=${1:-gpurun_out/r04}; mkdir -p $out
CAVIAR_HIP_TRACE_ITERS=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > $out/i_trace.json 2> $out/i_trace.err || exit 1
grep "^it " $out/i_trace.err | tail -49
