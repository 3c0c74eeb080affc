#!/usr/bin/env python3
"""Generate golden vectors by executing the reference's own CAVIaR source.

BUILD-CONTAINER TOOLING ONLY (needs /root/reference; never runs on the GPU box, never imported by
the product).  Loads ``/root/reference/circuitmap/optimise/{caviar,pava}.py`` BY PATH (the
reference package itself cannot be imported: jax and pytorch_lightning are absent) with
``tests/golden/jax_standin.py`` registered as `jax`, runs them on seeded inputs and stores
inputs + outputs as ``tests/golden/*.npz``.  Only data is written; no reference text is copied.

    python tests/golden/make_golden.py
"""
import contextlib
import importlib
import io
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import jax_standin  # noqa: E402
from circuitmap_amd.simulation import simulate  # noqa: E402
from oracle import threefry  # noqa: E402

REF_OPT = "/root/reference/circuitmap/optimise"


def load_reference():
    jax_standin.install()
    pkg = types.ModuleType("refopt")
    pkg.__path__ = [REF_OPT]
    sys.modules["refopt"] = pkg
    warnings.simplefilter("ignore")
    np.seterr(all="ignore")
    return importlib.import_module("refopt.caviar"), importlib.import_module("refopt.pava")


def default_priors(N):
    """reference circuitmap/model.py:24-31"""
    return dict(mu=np.zeros(N), beta=1e1 * np.ones(N), shape=1.0, rate=1e-1,
                phi=np.c_[1e-1 * np.ones(N), 5e0 * np.ones(N)],
                phi_cov=np.array([np.array([[1e-1, 0], [0, 1e0]]) for _ in range(N)]))


def run_caviar(ref, psc, stim, pri, **opts):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        out = ref.caviar(psc, stim, pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"],
                         pri["phi_cov"], **opts)
    names = ["mu", "beta", "lam", "shape", "rate", "phi", "phi_cov", "z"]
    res = {k: np.asarray(v, dtype=np.float64) for k, v in zip(names, out[:8])}
    if opts.get("save_histories"):
        for k, v in zip(names, out[9:17]):
            res["hist_" + k] = np.asarray(v, dtype=np.float64)
    return res


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def f32_exact(a):
    """Round inputs to float32-representable values so the fixture can store them compactly."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def case_cfg1(ref):
    """BASELINE configs[0]: N=50, K=200, T=900, 50 iterations, default options."""
    N, K = 50, 200
    sim = simulate(N=N, trials=K, seed=1001)
    psc, stim = f32_exact(sim["psc"]), sim["stim_matrix"]
    pri = default_priors(N)
    opts = dict(iters=50, seed=0, save_histories=True)
    res = run_caviar(ref, psc, stim, pri, **opts)
    orders, u = threefry.make_explicit_streams(0, 50, N, 100)
    save("caviar_cfg1", psc=psc.astype(np.float32), stim=stim.astype(np.float32), weights=sim["weights"],
         update_orders=orders, uniforms_seed=np.int64(0), iters=np.int64(50), **{"out_" + k: v for k, v in res.items()})


def case_cfg2(ref):
    """BASELINE configs[1]: N=300, K=1000, T=900, default options; the first 12 iterations (ungated full-size solves, the gating
    transition, the shrinking active set) + false-negative scan, with every per-iteration history.  lam (final and history) is
    stored on the support of the stimulus matrix only (it is identically zero elsewhere, caviar.py:216)."""
    N, K = 300, 1000
    sim = simulate(N=N, trials=K, seed=1300)
    psc, stim = f32_exact(sim["psc"]), sim["stim_matrix"]
    pri = default_priors(N)
    opts = dict(iters=12, seed=0, save_histories=True)
    res = run_caviar(ref, psc, stim, pri, **opts)
    sup = stim > 0
    assert not np.any(res["lam"][~sup]) and not np.any(res["hist_lam"][:, ~sup])
    res["lam"] = res["lam"][sup]
    res["hist_lam"] = res["hist_lam"][:, sup]
    res["hist_shape"], res["hist_rate"] = res["hist_shape"][:, 0].copy(), res["hist_rate"][:, 0].copy()   # scalars broadcast to [iters, K]
    save("caviar_cfg2", psc=psc.astype(np.float32), stim=stim.astype(np.float32), weights=sim["weights"],
         **{"opt_" + k: np.asarray(v) for k, v in opts.items()}, **{"out_" + k: v for k, v in res.items()})


def case_cfg3(ref, iters=8):
    """BASELINE configs[2] — the headline shape N=1000, K=2000, T=900 — default options: the first 8 iterations (three ungated
    full-size solves, the gating transition, the collapsing active set) + false-negative scan + final update_phi, with every
    per-iteration history.  lam (final and history) is stored on the support of the stimulus matrix only, as for cfg2; the
    stimulus matrix as its CSR triples' dense float32 form (compresses to a few hundred KB)."""
    N, K = 1000, 2000
    sim = simulate(N=N, trials=K, seed=2000)
    psc, stim = f32_exact(sim["psc"]), sim["stim_matrix"]
    pri = default_priors(N)
    opts = dict(iters=iters, seed=0, save_histories=True)
    res = run_caviar(ref, psc, stim, pri, **opts)
    sup = stim > 0
    assert not np.any(res["lam"][~sup]) and not np.any(res["hist_lam"][:, ~sup])
    res["lam"] = res["lam"][sup]
    res["hist_lam"] = res["hist_lam"][:, sup]
    res["hist_shape"], res["hist_rate"] = res["hist_shape"][:, 0].copy(), res["hist_rate"][:, 0].copy()
    save("caviar_cfg3", psc=psc.astype(np.float32), stim=stim.astype(np.float32), weights=sim["weights"],
         **{"opt_" + k: np.asarray(v) for k, v in opts.items()}, **{"out_" + k: v for k, v in res.items()})


def tiny_problem(seed, N=8, K=48, T=60, H=2, powers=(30.0, 50.0, 70.0)):
    rng = np.random.default_rng(seed)
    stim = np.zeros((N, K))
    for k in range(K):
        tars = rng.choice(N - 1, H, replace=False)     # cell N-1 is never targeted
        stim[tars, k] = powers[k % len(powers)]
    stim[N - 2, stim[N - 2] == powers[-1]] = powers[0]   # cell N-2 never sees the top power
    w = np.zeros(N)
    w[[0, 3]] = [25.0, 12.0]
    phi0, phi1 = 0.2, 9.0
    spk = (rng.random((N, K)) < 1 / (1 + np.exp(-(phi0 * stim - phi1)))) * (stim > 0)
    t = np.arange(T)
    kern = np.exp(-t / 20.0) - np.exp(-t / 5.0)
    kern /= np.trapezoid(kern)
    psc = (w @ spk)[:, None] * kern[None, :] + rng.normal(0, 5e-3, (K, T))
    spont = rng.choice(K, 4, replace=False)
    psc[spont] += 15.0 * kern
    psc[rng.choice(K, 5, replace=False)] = 0.0          # silent trials -> lam_mask false
    return f32_exact(psc), stim, w


def case_tiny(ref):
    """Edge cases: untargeted cell, empty power group, masked (all-zero) trials, non-default options,
    iters > max_backtrack_iters (z frozen from iteration 20), few MC samples, with and without fn_scan."""
    psc, stim, w = tiny_problem(7)
    N = stim.shape[0]
    pri = default_priors(N)
    pri["beta"] = 2e1 * np.ones(N)
    opts = dict(iters=25, num_mc_samples=7, seed=3, minimum_spike_count=2, msrmp=0.2, penalty=2.0,
                scale_factor=0.6, tol=0.01, spont_orthogonality=0.2, save_histories=True, fn_scan=True)
    res = run_caviar(ref, psc, stim, pri, **opts)
    orders, _ = threefry.make_explicit_streams(3, 25, N, 7)
    save("caviar_tiny", psc=psc.astype(np.float32), stim=stim.astype(np.float32), weights=w, beta_prior=pri["beta"],
         update_orders=orders, **{"opt_" + k: np.asarray(v) for k, v in opts.items()},
         **{"out_" + k: v for k, v in res.items()})
    opts2 = dict(opts, fn_scan=False, iters=6, delay_spont_est=3, y_xcorr_thresh=0.5)
    res2 = run_caviar(ref, psc, stim, pri, **opts2)
    save("caviar_tiny_noscan", psc=psc.astype(np.float32), stim=stim.astype(np.float32), beta_prior=pri["beta"],
         **{"opt_" + k: np.asarray(v) for k, v in opts2.items()}, **{"out_" + k: v for k, v in res2.items()})


def case_pava(pava):
    rng = np.random.default_rng(11)
    ys, outs = [], []
    for i in range(200):
        n = int(rng.integers(1, 7))
        y = np.round(rng.random(n), 1) if i % 2 else rng.random(n)   # rounded -> ties
        if i % 17 == 0:
            y[:] = 0.0
        ys.append(np.pad(y, (0, 6 - n), constant_values=np.nan))
        outs.append(np.pad(np.asarray(pava._isotonic_regression(y), dtype=np.float64), (0, 6 - n), constant_values=np.nan))
    save("pava", y=np.array(ys), out=np.array(outs))


def case_units(ref):
    """Per-phase known answers on random (not CAVIaR-trajectory) inputs, incl. sigmoid saturation in the
    phi objective (quirk 11) and a hand-built reconnect_spont_cells scenario (quirks 7, 8)."""
    jnp = sys.modules["jax.numpy"]
    rng = np.random.default_rng(5)
    N, K = 12, 80
    powers = np.array([40.0, 60.0, 80.0])
    stim = np.zeros((N, K))
    for k in range(K):
        stim[rng.choice(N, 3, replace=False), k] = powers[rng.integers(0, 3)]
    stim[2, :] = 0.0                              # untargeted row
    stim[5, stim[5] > 0] = 4000.0                 # saturating power on one row
    lam = rng.random((N, K)) * (stim > 0)
    lam[3, np.flatnonzero(stim[3] > 0)[:2]] = 1.0 # exact ones (reconnect leaves them)
    lam[4] = 0.0
    y = rng.normal(10, 5, K)
    mask = rng.random(K) > 0.1
    pri = default_priors(N)
    out = {}
    # block_update_mu / update_sigma
    mu, beta = ref.block_update_mu(jnp.array(y), None, None, jnp.array(lam), 3.0, 0.7, pri["mu"] + 0.5, pri["beta"], N)
    out["bum_mu"], out["bum_beta"] = np.asarray(mu), np.asarray(beta)
    sh, rt = ref.update_sigma(jnp.array(y), mu, beta, jnp.array(lam), 1.0, 0.1)
    out["sig_shape"], out["sig_rate"] = np.float64(sh), np.float64(rt)
    # update_phi
    (phi, phi_cov), key = ref.update_phi(jnp.array(lam), jnp.array(stim), jnp.array(pri["phi"]), jnp.array(pri["phi_cov"]),
                                         sys.modules["jax.random"].PRNGKey(9))
    out["phi"], out["phi_cov"], out["phi_key"] = np.asarray(phi), np.asarray(phi_cov), np.asarray(key)
    # soft threshold at several outer iterations
    for it in (0, 5, 19, 20):
        z = ref.estimate_spont_act_soft_thresh((jnp.array(y), mu, jnp.array(lam), it, 20, float(np.sum(y)), jnp.zeros(K) + 0.123,
                                                5.0, jnp.array(mask), 0.75, 0.05, 0.1))[6]
        out["z_it%d" % it] = np.asarray(z, dtype=np.float64)
    # reconnect_spont_cells
    mu_r = np.asarray(mu).copy()
    mu_r[[1, 2, 4, 7]] = 0.0
    z_r = np.zeros(K)
    t7 = np.flatnonzero(stim[7] > 0)
    z_r[t7[: max(4, len(t7) // 2)]] = rng.uniform(5, 9, max(4, len(t7) // 2))
    z_r[np.flatnonzero(stim[1] > 0)[:1]] = 3.0
    with contextlib.redirect_stdout(io.StringIO()):
        rm, rb, rl, rz = ref.reconnect_spont_cells(jnp.array(y), stim, jnp.array(lam), jnp.array(mu_r), jnp.array(np.asarray(beta)),
                                                   jnp.array(z_r), minimax_spk_prob=0.3, minimum_spike_count=3)
    out.update(rec_mu_in=mu_r, rec_z_in=z_r, rec_mu=np.asarray(rm), rec_beta=np.asarray(rb), rec_lam=np.asarray(rl), rec_z=np.asarray(rz))
    # update_lam: one Gauss-Seidel sweep, ungated (it <= delay_spont_est) and gated, from the phi posterior above
    jr = sys.modules["jax.random"]
    powers = np.unique(stim)[1:]
    y_s = y * 0.02                                   # residual scale at which the sigmoid posterior is not saturated
    for tag, it in (("ungated", 0), ("gated", 5)):
        key = jr.PRNGKey(3)
        lam_new, key_out = ref.update_lam(jnp.array(y_s), jnp.array(stim), mu * 0.02, beta * 1e-3, jnp.array(lam), 3.0, 0.7, phi, phi_cov,
                                          jnp.array(mask), key, 7, N, jnp.array(powers), 2.0, 0.35, it, 1)
        out["ul_lam_" + tag] = np.asarray(lam_new, dtype=np.float64)
        out["ul_key_" + tag] = np.asarray(key_out)
    out["ul_key_in"] = np.asarray(jr.PRNGKey(3))
    out["ul_y"], out["ul_mu"], out["ul_beta"] = y_s, np.asarray(mu) * 0.02, np.asarray(beta) * 1e-3
    save("units", stim=stim, lam=lam, y=y, mask=mask, mu_prior=pri["mu"] + 0.5, **out)


def main():
    ref, pava = load_reference()
    if len(sys.argv) > 1:                       # e.g. `make_golden.py cfg3`: one case only (cfg3 takes ~10 CPU-minutes)
        for name in sys.argv[1:]:
            fn = globals()["case_" + name]
            fn(pava if name == "pava" else ref)
        return
    case_pava(pava)
    case_units(ref)
    case_tiny(ref)
    case_cfg1(ref)
    case_cfg2(ref)
    case_cfg3(ref)


if __name__ == "__main__":
    main()
