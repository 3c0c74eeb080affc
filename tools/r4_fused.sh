#!/bin/bash
# the chain tables from block 0 of k_iter_mid + packing in the pre-sweep, against the separate launches (CAVIAR_HIP_NO_FUSED_PACK)
bash tools/r4prof.sh fused > /dev/null || exit 1
CAVIAR_HIP_NO_FUSED_PACK=1 bash tools/r4prof.sh unfused > /dev/null || exit 1
for t in fused unfused; do echo "== $t: $(python tools/bench_summary.py gpurun_out/r4prof/${t}_bench.json)"; grep -E "k_presweep|k_seq_pack|k_iter_mid|k_sweep_par|k_stats|k_mc_dead" gpurun_out/r4prof/${t}_kernel_stats.csv | cut -c1-110; done
for i in 1 2; do
python bench.py --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > gpurun_out/r4prof/f_$i.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/f_$i.json
CAVIAR_HIP_NO_FUSED_PACK=1 python bench.py --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > gpurun_out/r4prof/u_$i.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/u_$i.json
done
python bench.py --no-cpu-baseline --circuits-per-gpu 1 --batch-leg 0 > gpurun_out/r4prof/f_single.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/f_single.json
CAVIAR_HIP_NO_FUSED_PACK=1 python bench.py --no-cpu-baseline --circuits-per-gpu 1 --batch-leg 0 > gpurun_out/r4prof/u_single.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r4prof/u_single.json
