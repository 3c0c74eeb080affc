#!/bin/bash
mkdir -p gpurun_out/r4prof
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "chain_tables or ticket or many_per_xcd or bit_identical or long_rows or golden or bench_sizes or edge" > gpurun_out/r4prof/t_rank.txt 2>&1; tail -4 gpurun_out/r4prof/t_rank.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > gpurun_out/r4prof/rk_$i.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/rk_$i.json
CAVIAR_HIP_NO_COLUMN_RANKS=1 python bench.py --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > gpurun_out/r4prof/nrk_$i.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/nrk_$i.json
done
python bench.py --no-cpu-baseline --circuits-per-gpu 1 --batch-leg 0 > gpurun_out/r4prof/rk_single.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/rk_single.json
CAVIAR_HIP_NO_COLUMN_RANKS=1 python bench.py --no-cpu-baseline --circuits-per-gpu 1 --batch-leg 0 > gpurun_out/r4prof/nrk_single.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/nrk_single.json
