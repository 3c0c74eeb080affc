#!/bin/bash
# Sweep of circuits per step (B) and lock-step batches per step (G streams) on the default bench workload.
# usage: tools/stream_sweep.sh OUTDIR "B:G B:G ..."
out=${1:-gpurun_out/streams}
mkdir -p "$out"
for bg in ${2:-64:1 64:2 64:4 128:1 128:4 128:8}; do
  B=${bg%%:*}; G=${bg##*:}
  timeout -k 10 240 python bench.py --circuits-per-gpu $B --batch-streams $G --no-single-leg --streams-leg 0 --no-cpu-baseline --steps 5 --warmup 2 > "$out/b${B}_g${G}.json" 2> "$out/b${B}_g${G}.err" || exit 1
  python tools/bench_summary.py "$out/b${B}_g${G}.json"
done
