import sys, numpy as np
import os
ROOT=os.environ.get('ADV_ROOT','/root/repo')
sys.path.insert(0,ROOT); sys.path.insert(0,'/root/repo/tests')
import torch
from child_fit import priors_of
from circuitmap_amd.simulation import simulate_device
from circuitmap_amd.optimise import caviar
from oracle import caviar_oracle as co
from helpers import prior_args, OUT_NAMES, rel_err
specs=[]
for S in (3, 7):
    for delay in (0, 5):
        for phi, cov in ((None, None), ([0.5, 0.5], [0.5, 0.5])):
            for msrmp in (0.05, 0.2, 0.4, 0.7):
                specs.append(dict(N=120, K=400, B=8, seed0=6000 + len(specs), batch=True, beta_seed=17 + len(specs), phi=phi, phi_cov_diag=cov,
                                  opts=dict(iters=9, seed=len(specs), num_mc_samples=S, delay_spont_est=delay, msrmp=msrmp, minimum_spike_count=2, save_histories=True)))
for si in [int(x) for x in os.environ.get("ADV_CFGS","13").split(",")]:
    spec=specs[si]
    pri=priors_of(spec)
    d=simulate_device(N=120,T=900,H=10,trials=400,seed=spec["seed0"],full=False)
    res=caviar(d["psc"], d["stim_matrix"], *prior_args(pri), **spec["opts"])
    ref=co.caviar(d["psc"], d["stim_matrix"], *prior_args(pri), literal=False, **spec["opts"])
    errs={nm: rel_err(res[i], ref[i]) for i,nm in enumerate(OUT_NAMES)}
    ah=np.any(res[11]!=0,axis=2); ar=np.any(ref[11]!=0,axis=2)
    rev=sum(int(np.sum(~ar[it-1]&ar[it])) for it in range(1,ar.shape[0]))
    revh=sum(int(np.sum(~ah[it-1]&ah[it])) for it in range(1,ah.shape[0]))
    big={k:"%.1e"%v for k,v in errs.items() if v>1e-6}
    print(si, "hip revivals", revh, big, spec["opts"]["msrmp"], spec["phi"], "max err %.2e"%max(errs.values()), "alive equal", np.array_equal(ah,ar), "oracle revivals", rev, "alive per it", ar.sum(axis=1))
