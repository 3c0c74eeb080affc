"""per-seed single-fit statistics at cfg3: iterations that needed the tiled solve, fit time (ms)"""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from circuitmap_amd import _lib
from circuitmap_amd.optimise.caviar import Problem
from circuitmap_amd.simulation import simulate_device
from tests.helpers import default_priors, prior_args
N, K, T = 1000, 2000, 900
ctx = _lib.Context(0)
pri = default_priors(N)
for seed in range(2000, 2048):
    d = simulate_device(N=N, trials=K, seed=seed, keep_on_device=True, ctx=ctx)
    prob = Problem(ctx, None, None, device_ptrs=(N, K, T, d["psc_ptr"], d["stim_ptr"]))
    ctx.free(d["psc_ptr"]); ctx.free(d["stim_ptr"])
    prob.run(*prior_args(pri), iters=50, seed=0)
    tm = prob.run(*prior_args(pri), iters=50, seed=0, timing=1).as_dict()
    print("seed %d: total %.2f ms  fact %.2f ms  tiled launches %d  sum_active %d last %d" % (seed, tm["total_ms"], tm["fact_ms"], tm["fact_launches"], tm["sum_active"], tm["last_active"]))
    prob.close()
