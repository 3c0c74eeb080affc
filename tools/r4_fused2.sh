#!/bin/bash
for i in 1 2; do
python bench.py --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > gpurun_out/r4prof/f_$i.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/f_$i.json
CAVIAR_HIP_NO_FUSED_PACK=1 python bench.py --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > gpurun_out/r4prof/u_$i.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/u_$i.json
done
