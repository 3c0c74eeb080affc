set -x
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_bench_shapes or residue or adversarial or rerun or long_rows or rare or ticket or 3600 or phase_hooks or unit_phases" > gpurun_out/r4a/pytest1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4a/pytest1.log
tail -15 gpurun_out/r4a/pytest1.log
python bench.py --workload demo --no-cpu-baseline > gpurun_out/r4a/bench_demo.json 2> gpurun_out/r4a/bench_demo.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r4a/bench_demo.json
