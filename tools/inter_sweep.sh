#!/bin/bash
# usage: tools/inter_sweep.sh OUTDIR "B1 B2 ..."   -- one lock-step batch per step / the library's two interleaved halves, at each batch size
out=$1; mkdir -p $out
for B in $2; do
  for mode in ${MODES:-onebatch interleave}; do
    case $mode in
      onebatch) flag="";;
      interleave) flag="--interleave";;
    esac
    timeout -k 10 400 python bench.py $flag --circuits-per-gpu $B --steps 8 --warmup 2 --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > $out/b_${B}_$mode.json 2>$out/b_${B}_$mode.err || { echo "FAILED B=$B $mode"; tail -3 $out/b_${B}_$mode.err; exit 1; }
    python - $out/b_${B}_$mode.json $B $mode <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("B=%s %-8s %8.0f it/s %7.2f ms/step  equal=%s  dom=%s avg %.3f ms achieved %.2f frac %.3f" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d.get("batch_equals_single"), r.get("kernel","")[:14], r.get("avg_launch_ms",0), r["achieved"], r["frac"]))
PY
  done
done
