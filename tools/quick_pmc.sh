#!/bin/bash
# one PMC pass of a bench invocation.  usage: tools/quick_pmc.sh <outdir> <tag> "<counters>" -- <bench args...>
out=$1; tag=$2; ctrs=$3; shift 3
[ "$1" = "--" ] && shift
R=$(pwd); mkdir -p "$R/$out"; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/qc_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/qc_$tag -o p -- python3 "$R/bench.py" "$@" > /dev/null 2> "$R/$out/${tag}.err" || { echo "pmc pass failed"; tail -5 "$R/$out/${tag}.err"; exit 1; }
python3 "$R/tools/rocpd_pmc.py" /tmp/qc_$tag/p_results.db > "$R/$out/${tag}_pmc.csv"
cd "$R"
