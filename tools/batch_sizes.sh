#!/bin/bash
# lock-step batch sizes around the default (cfg3): one JSON line each
out=${1:-gpurun_out/r04/bsz}; mkdir -p $out
for B in ${2:-128 192 256 320 384 512}; do
  timeout -k 10 300 python bench.py --circuits-per-gpu $B --no-cpu-baseline --no-single-leg --no-end-to-end --steps 6 --warmup 2 > $out/B$B.json 2> $out/B$B.err || { echo "B=$B failed"; tail -3 $out/B$B.err; continue; }
  echo "B=$B $(python tools/bench_summary.py $out/B$B.json) | interleaved $(python -c "import json;d=json.load(open('$out/B$B.json'));print(round(d.get('interleaved',{}).get('value',0)))")"
done
