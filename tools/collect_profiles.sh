#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: kernel trace and separate PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy, SQ mix) of the
# default bench workload (tag a), of one circuit at a time (tag b), of BASELINE configs[4] as a batch (tag c: cfg5, N=5000 / K=10000, 32 circuits
# per step) and of the reference's published demo workload (tag d).  Run from the repo root on the GPU box; the third argument picks the legs (default "a b c"):
#     tools/collect_profiles.sh gpurun_out/r03/prof r03 "a c"
# Writes <out>/<round>_{a,b}_* CSV summaries + the bench JSON lines the passes printed.
set -o pipefail
out=${1:-gpurun_out/prof}; tag=${2:-rNN}
R=$(pwd)
mkdir -p "$R/$out"
export TMPDIR=/tmp
cd /tmp
A_ARGS="--circuits-per-gpu 256 --steps 3 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end"     # cfg3, 256 circuits per step (the default until round 4: the per-kernel analysis of DESIGN section 5 is on this leg)
E_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end"                              # the default bench workload: cfg3, 512 circuits per step
B_ARGS="--circuits-per-gpu 1 --batch-leg 0 --steps 3 --warmup 1 --no-cpu-baseline"
C_ARGS="--workload cfg5 --circuits-per-gpu 32 --steps 2 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0"     # BASELINE configs[4] as a lock-step batch (round 3: one circuit per step)
D_ARGS="--workload demo --steps 3 --warmup 1 --no-cpu-baseline --no-single-leg --streams-leg 0 --no-end-to-end"            # the reference's published workload (N = 200, K = 3600)
legs=${3:-e a b c d}
pass() {   # name, leg args, rocprof args...
  name=$1; legargs=$2; shift 2
  rm -rf /tmp/prof_$name
  timeout -k 10 400 rocprofv3 --kernel-trace "$@" -d /tmp/prof_$name -o p -- python3 "$R/bench.py" $legargs > "$R/$out/${tag}_${name}_bench.json" 2> "$R/$out/${tag}_${name}.err" || { echo "pass $name failed"; tail -5 "$R/$out/${tag}_${name}.err"; return 1; }
  echo "pass $name done"
}
for leg in $legs; do
  case $leg in a) args="$A_ARGS";; b) args="$B_ARGS";; c) args="$C_ARGS";; d) args="$D_ARGS";; e) args="$E_ARGS";; *) echo "unknown leg $leg"; exit 1;; esac
  pass ${leg}_trace "$args" --stats && python3 "$R/tools/rocpd_stats.py" /tmp/prof_${leg}_trace/p_results.db > "$R/$out/${tag}_${leg}_kernel_stats.csv" || exit 1
  pass ${leg}_fetch "$args" --pmc FETCH_SIZE && python3 "$R/tools/rocpd_pmc.py" /tmp/prof_${leg}_fetch/p_results.db > "$R/$out/${tag}_${leg}_pmc_fetch_size.csv" || exit 1
  pass ${leg}_write "$args" --pmc WRITE_SIZE && python3 "$R/tools/rocpd_pmc.py" /tmp/prof_${leg}_write/p_results.db > "$R/$out/${tag}_${leg}_pmc_write_size.csv" || exit 1
  pass ${leg}_mfma "$args" --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE && python3 "$R/tools/rocpd_pmc.py" /tmp/prof_${leg}_mfma/p_results.db > "$R/$out/${tag}_${leg}_pmc_mfma_busy.csv" || exit 1
  pass ${leg}_sq "$args" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD && python3 "$R/tools/rocpd_pmc.py" /tmp/prof_${leg}_sq/p_results.db > "$R/$out/${tag}_${leg}_pmc_sq.csv" || exit 1
done
cd "$R"
rm -f "$out"/*_fetch_bench.json "$out"/*_write_bench.json "$out"/*_mfma_bench.json "$out"/*_sq_bench.json
ls -la "$out"
