#!/bin/bash
# per-role durations of k_presweep (rows / columns / scan + pack) at the default step: CAVIAR_HIP_SPLIT_PRESWEEP=1 under rocprofv3
out=gpurun_out/r04/roles; mkdir -p $out
R=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/roles_p
export CAVIAR_HIP_SPLIT_PRESWEEP=1
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/roles_p -o p -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end > "$R/$out/bench.json" 2> "$R/$out/err.txt" || { tail -3 "$R/$out/err.txt"; exit 1; }
python3 "$R/tools/presweep_roles.py" /tmp/roles_p/p_results.db
