set -x
out=gpurun_out/r4b; mkdir -p $out
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/qp_s
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/qp_s -o p -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > "$R/$out/s_bench.json" 2> "$R/$out/s.err" || { tail -5 "$R/$out/s.err"; exit 1; }
python3 "$R/tools/rocpd_stats.py" /tmp/qp_s/p_results.db > "$R/$out/s_kernel_stats.csv"
python3 "$R/tools/rocpd_seq.py" /tmp/qp_s/p_results.db 900 > "$R/$out/s_seq.txt"
cd "$R"
CAVIAR_HIP_NO_PAR_SWEEP=1 python bench.py --workload demo --no-cpu-baseline --no-end-to-end --streams-leg 0 --steps 5 > $out/demo_seq.json 2> $out/demo_seq.err
python tools/solve_bench.py 1000 256 3 > $out/solve_bench.txt 2>&1
python tools/solve_bench.py 1000 128 3 >> $out/solve_bench.txt 2>&1
cat $out/solve_bench.txt
