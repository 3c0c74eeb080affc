#!/usr/bin/env python3
"""one-line summaries of bench.py JSON lines: python tools/bench_summary.py gpurun_out/x.json ..."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e)
        continue
    r = d["roofline"]; p = d["phase_ms_per_step"]
    print("%s: %.0f it/s  %.2f ms/step  dom %s %.1f us | mid %.2f seq %.2f fact %.2f solve %.2f total %.2f fn %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"] * 1e3, p["mid_ms"], p["sweep_seq_ms"], p["fact_ms"],
        p["solve_ms"], p["total_ms"], p["fn_scan_ms"]))
    if "batch_throughput" in d:
        b = d["batch_throughput"]
        print("    batch leg B=%d: %.0f it/s (%.2fx) %.2f ms/step mid %.1f us" % (b["circuits_per_gpu"], b["value"], b["x_single_circuit"], b["ms_per_step"], b["k_iter_mid_avg_launch_ms"] * 1e3))
