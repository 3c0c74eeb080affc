#!/usr/bin/env python3
"""Wall time of a complete Model.fit(psc, stim, method='caviar') from host arrays (PCIe-inclusive; never the bench's `value`):
    python tools/model_fit_time.py [N] [K] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime per process: torch first)
from circuitmap_amd.model import Model  # noqa: E402
from circuitmap_amd.simulation import simulate  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sim = simulate(N=N, T=900, H=10, trials=K, seed=2000)
psc, stim = np.ascontiguousarray(sim["psc"], dtype=float), np.ascontiguousarray(sim["stim_matrix"], dtype=float)
ts = []
for r in range(reps + 2):
    m = Model(N)
    t0 = time.perf_counter()
    m.fit(psc, stim, method="caviar", fit_options=dict(iters=50, seed=0, save_histories=False))
    ts.append((time.perf_counter() - t0) * 1e3)
print("Model.fit N=%d K=%d: median %.2f ms, best %.2f ms over %d fits (first two discarded: %.1f, %.1f ms)" % (N, K, float(np.median(ts[2:])), min(ts[2:]), reps, ts[0], ts[1]))
