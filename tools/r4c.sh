out=gpurun_out/r4c; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unit_phases or precision_build or golden or cholesky_families or bit_identical or oracle_bench_sizes or rare or long_rows or residue" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
python bench.py --no-cpu-baseline --no-end-to-end > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
python tools/bench_summary.py $out/bench.json 2>/dev/null || tail -c 600 $out/bench.json
