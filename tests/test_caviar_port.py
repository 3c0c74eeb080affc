"""The compiled CPU restatement (oracle/caviar_port.cpp: C++17 / OpenMP, what bench.py times as `cpu_baseline`) against the fixtures recorded
from the reference's own source and against the NumPy oracle.  CPU only; test infrastructure checking test infrastructure -- the product
never loads it (asserted below)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import OUT_NAMES, load_case, prior_args, rel_err
from oracle import caviar_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cp():
    from oracle import caviar_port
    if not os.path.exists(caviar_port.LIB_PATH):        # (the driver runs build() first; a bare checkout builds it here)
        subprocess.run(["g++", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-std=c++17", os.path.join(ROOT, "oracle", "caviar_port.cpp"),
                        "-o", caviar_port.LIB_PATH], check=True)
    return caviar_port


@pytest.mark.parametrize("name,tol", [("caviar_tiny", 1e-8), ("caviar_tiny_noscan", 1e-8), ("caviar_cfg1", 1e-8), ("caviar_cfg2", 1e-6), ("caviar_cfg3", 1e-6)])
def test_port_vs_reference_source_fixtures(cp, name, tol):
    """Final outputs (mu, beta, lam, shape, rate, phi, phi_cov, z) of the compiled port against tests/golden/*.npz (reference caviar.py:20-100 run
    under the jax stand-in): 1e-8 where the run is long enough to have left the ill-conditioned early solves, 1e-6 for the 12- and 8-iteration
    fixtures at BASELINE configs[1] and configs[2] (the headline shape), cf. tests/test_oracle_golden.py."""
    g, psc, stim, pri, opts = load_case(name)
    opts = {k: v for k, v in opts.items() if k != "save_histories"}
    for threads in (1, 4):
        res = cp.caviar(psc, stim, *prior_args(pri), n_threads=threads, **opts)
        sup = stim > 0
        for i, nm in enumerate(OUT_NAMES):
            got = np.asarray(res[i], dtype=float)
            want = g["out_" + nm]
            if nm == "lam" and want.ndim == 1:
                assert not np.any(got[~sup])
                got = got[sup]
            assert rel_err(got, want) < tol, (name, nm, threads, rel_err(got, want))


def test_port_vs_oracle_nondefault_options(cp):
    """Non-default options and per-neuron priors on a seeded synthetic circuit: the port follows the NumPy oracle's running-residual formulation."""
    from circuitmap_amd.simulation import simulate
    from helpers import default_priors
    sim = simulate(N=60, trials=240, seed=77)
    pri = default_priors(60)
    rng = np.random.default_rng(3)
    pri["beta"] = rng.uniform(2.0, 20.0, 60)
    pri["mu"] = rng.normal(0.0, 0.05, 60)
    opts = dict(iters=24, num_mc_samples=9, seed=5, minimum_spike_count=2, msrmp=0.2, penalty=3.0, scale_factor=0.6, tol=0.02, spont_orthogonality=0.2,
                delay_spont_est=3, y_xcorr_thresh=0.02, max_backtrack_iters=15, fn_scan=True)
    res = cp.caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), **opts)
    ref = co.caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), literal=False, **opts)
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(res[i], ref[i]) < 1e-7, (nm, rel_err(res[i], ref[i]))


def test_product_never_touches_the_port():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.build() / smoke() may know about oracle/: no file of the product package names it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "circuitmap_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "caviar_port" not in txt and "libcaviar_port" not in txt, os.path.join(dirpath, f)
