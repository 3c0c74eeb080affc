import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from circuitmap_amd import _lib
from oracle import caviar_oracle as co
from helpers import rel_err
ctx=_lib.default_context(0)
rng=np.random.default_rng(1)
for trial,(pm,pv) in enumerate([((0.5,0.5),(0.5,0.5)), ((0.1,5.0),(0.1,1.0)), ((0.05,0.3),(2.0,2.0)), ((1.0,0.2),(0.05,0.05))]):
    N,K=160,400
    powers=np.array([45.0,55.0,65.0])
    stim=np.zeros((N,K))
    for k in range(K):
        stim[rng.choice(N,10,replace=False),k]=powers[rng.integers(0,3)]
    lam=rng.random((N,K))*(stim>0)*(rng.random((N,K))<0.6)
    lam[rng.random((N,K))<0.02]=1.0; lam*= (stim>0)
    lam[:20]=0.0
    lam[20:40]*= (rng.random((20,K))<0.1)
    phi_prior=np.tile(np.array(pm),(N,1)); cov=np.array([np.diag(pv) for _ in range(N)])
    phi,pc=np.empty((N,2)),np.empty((N,2,2))
    _lib.check(ctx.lib.cm_k_update_phi(ctx.handle,N,K,_lib.dptr(stim),_lib.dptr(_lib.f64(lam)),_lib.dptr(phi_prior),_lib.dptr(cov),_lib.dptr(phi),_lib.dptr(pc)))
    rphi,rcov=co.update_phi(lam,stim,phi_prior,cov,powers,literal=False)
    e=np.abs(phi-rphi).max(axis=1)/np.abs(rphi).max()
    print(trial, pm, pv, "phi err %.2e cov err %.2e"%(rel_err(phi,rphi),rel_err(pc,rcov)), "rows>1e-8:", np.flatnonzero(e>1e-8)[:10])
