# usage: tools/r4prof.sh <tag> [bench args...]: one timed step under rocprofv3 --kernel-trace: per-kernel stats + dispatch sequence of the last 1300 kernels
tag=$1; shift
out=gpurun_out/r4prof; mkdir -p $out
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/qp_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/qp_$tag -o p -- python3 "$R/bench.py" --steps ${PROF_STEPS:-1} --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end "$@" > "$R/$out/${tag}_bench.json" 2> "$R/$out/${tag}.err" || { tail -5 "$R/$out/${tag}.err"; exit 1; }
python3 "$R/tools/rocpd_stats.py" /tmp/qp_$tag/p_results.db > "$R/$out/${tag}_kernel_stats.csv"
python3 "$R/tools/rocpd_seq.py" /tmp/qp_$tag/p_results.db ${SEQ_N:-1300} > "$R/$out/${tag}_seq.txt"
cd "$R"; head -30 $out/${tag}_kernel_stats.csv | cut -c1-120
