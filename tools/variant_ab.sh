#!/bin/bash
# A/B of library variants built by tools/build_variant.sh: tools/variant_ab.sh "<name> <name> ..." [bench args]  (the default build first)
names=$1; shift
out=gpurun_out/r04/ab; mkdir -p $out
run() { tag=$1; shift; "$@" python bench.py --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > $out/$tag.json 2> $out/$tag.err && echo "$tag: $(python tools/bench_summary.py $out/$tag.json)"; }
for rep in 1 2; do
  run default_$rep env
  for n in $names; do run ${n}_$rep env CAVIAR_HIP_LIB=$(pwd)/build/variants/lib$n.so; done
done
