#!/bin/bash
# The bench lines behind BASELINE.md section 6: default (with cpu_baseline), cfg2 (one circuit / batch), cfg3 one circuit, cfg5 (one circuit and as
# lock-step batches of 32 / 64: BASELINE configs[4] as a throughput stress), the reference's published demo workload.
out=${1:-gpurun_out/lines}; tag=${2:-rNN}
mkdir -p "$out"
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > "$out/${tag}_$name.json" 2> "$out/${tag}_$name.err" || { echo "$name failed"; tail -3 "$out/${tag}_$name.err"; return 1; }; python tools/bench_summary.py "$out/${tag}_$name.json"; }
run default_bench_with_cpu_baseline &&
run bench_cfg3_B256 --circuits-per-gpu 256 --no-cpu-baseline &&
run bench_cfg3_B1 --circuits-per-gpu 1 --no-cpu-baseline &&
run bench_cfg2_B1 --workload cfg2 --circuits-per-gpu 1 --no-cpu-baseline &&
run bench_cfg2_B256 --workload cfg2 --no-cpu-baseline &&
run bench_demo --workload demo &&
run bench_cfg5 --workload cfg5 --steps 4 --warmup 1 --no-cpu-baseline &&
run bench_cfg5_B32 --workload cfg5 --circuits-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0 &&
run bench_cfg5_B64 --workload cfg5 --circuits-per-gpu 64 --steps 3 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0
