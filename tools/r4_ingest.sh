#!/bin/bash
# where the time of cm_caviar_fit_batch(_ex) from host arrays goes (CAVIAR_HIP_FIT_BATCH_TIMES), staged and plain uploads
out=${1:-gpurun_out/r04}; mkdir -p $out
export CAVIAR_HIP_FIT_BATCH_TIMES=1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-single-leg --streams-leg 0 > $out/ing_staged.json 2> $out/ing_staged.err || exit 1
CAVIAR_HIP_PLAIN_UPLOAD=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-single-leg --streams-leg 0 > $out/ing_plain.json 2> $out/ing_plain.err || exit 1
grep "cm_caviar_fit_batch device" $out/ing_staged.err; echo; grep "cm_caviar_fit_batch device" $out/ing_plain.err
