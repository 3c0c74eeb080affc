"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT_NAMES = ["mu", "beta", "lam", "shape", "rate", "phi", "phi_cov", "z"]


def default_priors(N, beta=None):
    """Model defaults, reference circuitmap/model.py:24-31."""
    return dict(mu=np.zeros(N), beta=(1e1 * np.ones(N) if beta is None else np.asarray(beta, dtype=float)),
                shape=1.0, rate=1e-1, phi=np.c_[1e-1 * np.ones(N), 5e0 * np.ones(N)],
                phi_cov=np.array([np.array([[1e-1, 0], [0, 1e0]]) for _ in range(N)]))


def load_case(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    psc = g["psc"].astype(np.float64)
    stim = g["stim"].astype(np.float64)
    pri = default_priors(stim.shape[0], g["beta_prior"] if "beta_prior" in g.files else None)
    if name == "caviar_cfg1":
        opts = dict(iters=50, seed=0, save_histories=True)
    else:
        opts = {k[4:]: g[k].item() for k in g.files if k.startswith("opt_")}
    return g, psc, stim, pri, opts


def prior_args(pri):
    return (pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"], pri["phi_cov"])


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    scale = max(np.nanmax(np.abs(b)) if b.size else 0.0, 1e-300)
    return (np.nanmax(np.abs(a - b)) if a.size else 0.0) / scale
