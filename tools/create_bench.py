#!/usr/bin/env python3
"""cm_problem_create cost: device-resident inputs (CM_MEM_DEVICE) and host arrays, cfg3 shape.   python tools/create_bench.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from circuitmap_amd import _lib  # noqa: E402
from circuitmap_amd.optimise.caviar import Problem  # noqa: E402
from circuitmap_amd.simulation import simulate_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, K, T = 1000, 2000, 900
ctx = _lib.Context(0)
d = simulate_device(N=N, T=T, H=10, trials=K, seed=2000, keep_on_device=True, ctx=ctx)
psc, stim = ctx.download(d["psc_ptr"], np.empty((K, T))), ctx.download(d["stim_ptr"], np.empty((N, K)))
for label, mk in (("device-resident", lambda: Problem(ctx, None, None, device_ptrs=(N, K, T, d["psc_ptr"], d["stim_ptr"]))), ("host arrays", lambda: Problem(ctx, psc, stim))):
    q = mk(); q.close()
    ctx.synchronize()
    t0 = time.perf_counter()
    ps = [mk() for _ in range(n)]
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    t1 = time.perf_counter()
    for q in ps:
        q.close()
    print("%-16s create %.3f ms per circuit, destroy %.3f ms" % (label, dt * 1e3, (time.perf_counter() - t1) / n * 1e3))
