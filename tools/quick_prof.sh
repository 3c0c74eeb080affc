#!/bin/bash
# (QP_SCRIPT=tools/solve_bench.py profiles another script of the repo instead of bench.py)
# quick look: kernel-trace stats (+ optional MFMA-busy pass) of a bench invocation.  usage: tools/quick_prof.sh <outdir> <tag> [mfma] -- <bench args...>
out=$1; tag=$2; shift 2
mfma=0; if [ "$1" = mfma ]; then mfma=1; shift; fi
[ "$1" = "--" ] && shift
R=$(pwd); mkdir -p "$R/$out"; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/qp_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/qp_$tag -o p -- python3 "$R/${QP_SCRIPT:-bench.py}" "$@" > "$R/$out/${tag}_bench.json" 2> "$R/$out/${tag}.err" || { echo "trace pass failed"; tail -5 "$R/$out/${tag}.err"; exit 1; }
python3 "$R/tools/rocpd_stats.py" /tmp/qp_$tag/p_results.db > "$R/$out/${tag}_kernel_stats.csv"
if [ $mfma = 1 ]; then
  rm -rf /tmp/qp_${tag}_m
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/qp_${tag}_m -o p -- python3 "$R/${QP_SCRIPT:-bench.py}" "$@" > /dev/null 2> "$R/$out/${tag}_m.err" || { echo "mfma pass failed"; exit 1; }
  python3 "$R/tools/rocpd_pmc.py" /tmp/qp_${tag}_m/p_results.db > "$R/$out/${tag}_pmc_mfma_busy.csv"
fi
cd "$R"; head -25 "$out/${tag}_kernel_stats.csv"
