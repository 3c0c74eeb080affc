import sys, numpy as np
sys.path.insert(0, '.')
import torch
from circuitmap_amd import _lib
from circuitmap_amd.optimise.caviar import Problem
from circuitmap_amd.simulation import simulate_device
from tests.helpers import default_priors, prior_args
N,K,T=1000,2000,900
d = simulate_device(N=N, trials=K, seed=2000, keep_on_device=True)
ctx=d["ctx"]
prob = Problem(ctx, None, None, device_ptrs=(N,K,T,d["psc_ptr"],d["stim_ptr"]))
pri=default_priors(N)
prob.run(*prior_args(pri), iters=50, seed=0)
prob.run(*prior_args(pri), iters=50, seed=0, verbose=3)
