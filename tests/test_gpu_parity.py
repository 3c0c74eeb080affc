"""GPU parity: HIP path (through the C ABI) vs the CPU oracle and the golden vectors of the reference source."""
import ctypes as C

import numpy as np
import pytest

from helpers import GOLDEN, OUT_NAMES, default_priors, load_case, prior_args, rel_err

pytestmark = pytest.mark.gpu

# fp64 throughout on both sides.  Rounding-level differences (Cholesky vs LU inverse, summation order, libm) are
# transiently amplified by ~1e7 in the early ill-conditioned iterations (see test_oracle_golden), hence 1e-6 on
# per-iteration histories; final outputs are held to 1e-8 (north_star tolerance: 1e-4 relative on mu/lam).
TOL_FINAL = 1e-8
TOL_HIST = 1e-6


@pytest.fixture(scope="module")
def ctx():
    from circuitmap_amd import _lib
    return _lib.default_context(0)


def test_prologue_matches_oracle(ctx):
    from circuitmap_amd import _lib
    from oracle import caviar_oracle as co
    rng = np.random.default_rng(0)
    for K, T in [(1, 1), (3, 2), (7, 63), (5, 64), (9, 65), (33, 900), (200, 901)]:
        psc = rng.normal(0, 1, (K, T))
        y, ss = np.empty(K), np.empty(K)
        _lib.check(ctx.lib.cm_k_prologue(ctx.handle, K, T, _lib.dptr(psc), _lib.dptr(y), _lib.dptr(ss)))
        y0, mask0, *_ = co.prologue(psc, np.zeros((2, K)))
        assert np.allclose(y, y0, rtol=1e-12, atol=1e-12)
        assert np.allclose(ss, np.sum(psc * psc, axis=1), rtol=1e-12)


def test_rng_streams_match_threefry_oracle(ctx):
    from circuitmap_amd import _lib
    from oracle import threefry
    for seed, N, S, iters in [(0, 50, 100, 3), (3, 8, 7, 4), (12345678901, 1001, 5, 2), (1, 2000, 3, 1)]:
        orders, u = threefry.make_explicit_streams(seed, iters, N, S)
        for it in range(iters):
            o = np.empty(N, dtype=np.int32)
            uu = np.empty((N, S, 2))
            _lib.check(ctx.lib.cm_k_rng_streams(ctx.handle, seed, it, N, S, o.ctypes.data_as(C.POINTER(C.c_int32)), _lib.dptr(uu)))
            assert np.array_equal(o, orders[it])
            assert np.array_equal(uu, u[it])          # bit-exact: integer PRNG + exact bit-to-float conversion


def test_device_quantile_matches_scipy(ctx):
    """cm_ndtri (Wichura AS 241 on the device) against SciPy's ndtri -- what the oracle and the golden fixtures use for jax.scipy.special.ndtri
    of reference caviar.py:212: 1e-14 relative over uniform arguments, the far tails and the centre / tail boundaries; exact at the special values."""
    from circuitmap_amd import _lib
    from scipy.special import ndtri
    rng = np.random.default_rng(4)
    p = np.concatenate([rng.random(200000), 10.0 ** rng.uniform(-300, -1, 20000), 1 - 10.0 ** rng.uniform(-16, -1, 20000),
                        [0.075, 0.925, 0.5, 2.0 ** -52, 1 - 2.0 ** -53, 0.07499999999999999, 0.9250000000000002, 5e-324, 1.3887943864964021e-11]])
    out = np.empty_like(p)
    _lib.check(ctx.lib.cm_k_ndtri(ctx.handle, len(p), _lib.dptr(p), _lib.dptr(out)))
    ref = ndtri(p)
    big = np.abs(ref) > 1e-3
    assert np.max(np.abs(out - ref)[big] / np.abs(ref)[big]) < 1e-14 and np.max(np.abs(out - ref)[~big]) < 1e-17
    sp = np.array([0.0, 1.0, -0.1, 1.1, np.nan])
    o2 = np.empty_like(sp)
    _lib.check(ctx.lib.cm_k_ndtri(ctx.handle, len(sp), _lib.dptr(sp), _lib.dptr(o2)))
    assert o2[0] == -np.inf and o2[1] == np.inf and np.all(np.isnan(o2[2:]))


def test_unit_phases_vs_golden(ctx):
    from circuitmap_amd import _lib
    g = np.load(GOLDEN + "/units.npz")
    stim, lam, y = g["stim"], g["lam"], g["y"]
    N, K = stim.shape
    pri = default_priors(N)
    mu, beta = np.empty(N), np.empty(N)
    _lib.check(ctx.lib.cm_k_block_update_mu(ctx.handle, N, K, _lib.dptr(stim), _lib.dptr(lam), _lib.dptr(y), 3.0, 0.7,
                                            _lib.dptr(_lib.f64(g["mu_prior"])), _lib.dptr(pri["beta"]), _lib.dptr(mu), _lib.dptr(beta)))
    assert rel_err(mu, g["bum_mu"]) < 1e-11 and rel_err(beta, g["bum_beta"]) < 1e-11
    phi, cov = np.empty((N, 2)), np.empty((N, 2, 2))
    _lib.check(ctx.lib.cm_k_update_phi(ctx.handle, N, K, _lib.dptr(stim), _lib.dptr(lam), _lib.dptr(pri["phi"]), _lib.dptr(pri["phi_cov"]),
                                       _lib.dptr(phi), _lib.dptr(cov)))
    assert rel_err(phi, g["phi"]) < 1e-9 and rel_err(cov, g["phi_cov"]) < 1e-9
    # reconnect_spont_cells (host C++ on the compressed rows): integer / rational arithmetic -> bit-exact
    rl, rm, rb, rz = lam.copy(), g["rec_mu_in"].copy(), g["bum_beta"].copy(), g["rec_z_in"].copy()
    _lib.check(ctx.lib.cm_k_reconnect(ctx.handle, N, K, _lib.dptr(stim), _lib.dptr(rl), _lib.dptr(rm), _lib.dptr(rb), _lib.dptr(rz), 0.3, 3.0))
    assert np.array_equal(rm, g["rec_mu"]) and np.array_equal(rl, g["rec_lam"]) and np.array_equal(rz, g["rec_z"])
    assert rel_err(rb, g["rec_beta"]) < 1e-14


@pytest.mark.parametrize("case", ["long_rows", "wide_trials", "short"])
def test_block_update_mu_precision_build_paths(ctx, case):
    """cm_k_block_update_mu vs the oracle on shapes that drive the precision-matrix build (k_build_precision) through its other paths:
    rows of more than 64 trials (several 64-trial chunks per row), trials with more than 256 targeted neurons (a trial walked directly
    instead of staged: CM_BP_ITEMS), and the plain case; every active set is well above the in-LDS solver's cap, so the tiled
    factorisation (panel family, 2-5 tile rows) runs behind the build.  Some rows are all zero (they leave the active set)."""
    from circuitmap_amd import _lib
    from oracle import caviar_oracle as co
    N, K, H = {"long_rows": (96, 700, 12), "wide_trials": (400, 60, 300), "short": (150, 300, 6)}[case]
    rng = np.random.default_rng({"long_rows": 3, "wide_trials": 4, "short": 5}[case])
    stim = np.zeros((N, K))
    for k in range(K):
        stim[rng.choice(N, H, replace=False), k] = rng.choice([40.0, 55.0, 70.0])
    lam = rng.random((N, K)) * (stim > 0) * (rng.random((N, K)) < 0.7)
    lam[rng.choice(N, N // 6, replace=False)] = 0.0                   # rows outside the active set
    y = rng.gamma(2.0, 1.0, K)
    pri = default_priors(N)
    mu0 = _lib.f64(rng.normal(0.0, 0.1, N))
    if case == "long_rows":
        assert (stim > 0).sum(axis=1).max() > 64
    if case == "wide_trials":
        assert (stim > 0).sum(axis=0).max() > 256
    mu, beta = np.empty(N), np.empty(N)
    _lib.check(ctx.lib.cm_k_block_update_mu(ctx.handle, N, K, _lib.dptr(_lib.f64(stim)), _lib.dptr(_lib.f64(lam)), _lib.dptr(_lib.f64(y)), 3.0, 0.7,
                                            _lib.dptr(mu0), _lib.dptr(pri["beta"]), _lib.dptr(mu), _lib.dptr(beta)))
    rmu, rbeta = co.block_update_mu(y, lam, 3.0, 0.7, mu0, pri["beta"], literal=True)
    assert rel_err(mu, rmu) < 1e-9 and rel_err(beta, rbeta) < 1e-9, (case, rel_err(mu, rmu), rel_err(beta, rbeta))


@pytest.mark.parametrize("name", ["caviar_tiny", "caviar_tiny_noscan", "caviar_cfg1"])
def test_caviar_vs_golden(name):
    """Seeded threefry stream generated ON DEVICE; outputs and all per-iteration histories vs the reference-source fixtures."""
    from circuitmap_amd.optimise import caviar
    g, psc, stim, pri, opts = load_case(name)
    res = caviar(psc, stim, *prior_args(pri), **opts)
    assert res[8] is None
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(res[i], g["out_" + nm]) < TOL_FINAL, nm
        assert rel_err(res[9 + i], g["out_hist_" + nm]) < TOL_HIST, "hist " + nm


def test_explicit_streams_equal_seeded(ctx):
    from circuitmap_amd.optimise import caviar
    from oracle import threefry
    g, psc, stim, pri, opts = load_case("caviar_tiny")
    orders, u = threefry.make_explicit_streams(opts["seed"], opts["iters"], stim.shape[0], opts["num_mc_samples"])
    a = caviar(psc, stim, *prior_args(pri), **opts)
    b = caviar(psc, stim, *prior_args(pri), update_orders=orders, uniforms=u, **opts)
    for i, nm in enumerate(OUT_NAMES):
        assert np.array_equal(np.asarray(a[i]), np.asarray(b[i]), equal_nan=True), nm


def _sim_case(N, K, seed, **kw):
    from circuitmap_amd.simulation import simulate
    sim = simulate(N=N, trials=K, seed=seed, **kw)
    return sim, default_priors(N)


@pytest.mark.parametrize("N,K,iters", [(300, 1000, 50), (1000, 2000, 50), (2000, 4000, 3), (1000, 6000, 3)])
def test_caviar_vs_oracle_bench_sizes(N, K, iters):
    """BASELINE configs[1] and configs[2] (the headline shape), FULL trajectories: all 50 iterations + the false-negative scan
    + the final update_phi (reference caviar.py:79-98) against the running-residual oracle on identical inputs.  N = 2000
    exercises the two-round shuffle (N > 1625), the large-matrix Cholesky path (63 tiles) and K > 2048; K = 6000 takes the
    multi-wave chain with 8-bit tickets and 512-record buffers."""
    from circuitmap_amd.optimise import caviar
    from oracle import caviar_oracle as co
    sim, pri = _sim_case(N, K, 1000 + N)
    opts = dict(iters=iters, seed=0)
    res = caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), **opts)
    ref = co.caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), literal=False, **opts)
    # 50 iterations: the trajectory has long left the ill-conditioned early solves; the committed report (profiles/r0*_parity_report.txt)
    # measures 3e-8 worst case over all outputs -- held to 1e-7, so a regression of half an order of magnitude shows.  3 iterations end
    # inside the amplified phase (see TOL_HIST).
    tol = 1e-7 if iters == 50 else 1e-6
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(res[i], ref[i]) < tol, (nm, rel_err(res[i], ref[i]))
    if iters == 50:
        # the connected set the two sides report is the same set (gate decisions are discontinuous: a flipped gate would show here)
        assert np.array_equal(np.asarray(res[0]) != 0, np.asarray(ref[0]) != 0)
        assert np.array_equal(np.any(res[2] != 0, axis=1), np.any(ref[2] != 0, axis=1))


@pytest.mark.parametrize("N,K,seed", [(1000, 2000, 3100), (1000, 2000, 3101), (300, 1000, 3102), (200, 3600, 3103)])
def test_caviar_vs_compiled_port_full_runs(N, K, seed):
    """A third, independent witness at the headline shape: the compiled C++ restatement (oracle/caviar_port.cpp -- its own threefry, quantile,
    blocked Cholesky, sweep, Newton and false-negative scan; pinned by the reference-source fixtures in tests/test_caviar_port.py) on the same
    seeded inputs for all 50 iterations + fn_scan (caviar.py:79-98).  HIP against it: 1e-6 on every output (the port factorises and sums in
    its own orders: its own distance to the fixtures is 1e-6 at these shapes), the same connected set and the same rows with lam != 0."""
    from circuitmap_amd.optimise import caviar
    from oracle import caviar_port
    kw = dict(connection_prob=0.1) if K == 3600 else {}
    sim, pri = _sim_case(N, K, seed, **kw)
    opts = dict(iters=50, seed=1)
    res = caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), **opts)
    ref = caviar_port.caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), n_threads=8, **opts)
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(res[i], ref[i]) < 1e-6, (nm, rel_err(res[i], ref[i]))
    assert np.array_equal(np.asarray(res[0]) != 0, np.asarray(ref[0]) != 0)
    assert np.array_equal(np.any(res[2] != 0, axis=1), np.any(ref[2] != 0, axis=1))
    assert np.count_nonzero(res[0]) > 0


@pytest.mark.parametrize("seed", [7, 8])
def test_accuracy_anchor_notebook_shape(seed):
    """End-to-end accuracy anchor: the reference's demo recovers the synaptic weights with R^2 = 0.9546 at N = 200
    neurons, K ~ 3600 trials, H = 10, connection_prob = 0.1 (examples/demo_compressive_connectivity_mapping.ipynb:593; ~180 trials
    per cell).  Same shape here on seeded synthetic trial data, default fit options: R^2(weights, mu) >= 0.9 for the HIP path AND
    for the oracle, no false positives, and the two agree to parity tolerance."""
    from circuitmap_amd.optimise import caviar
    from oracle import caviar_oracle as co
    sim, pri = _sim_case(200, 3600, seed, H=10, connection_prob=0.1)
    w = sim["weights"]
    res = caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), iters=50, seed=0)
    ref = co.caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), literal=False, iters=50, seed=0)
    r2_hip = np.corrcoef(w, res[0])[0, 1] ** 2
    r2_ref = np.corrcoef(w, ref[0])[0, 1] ** 2
    assert r2_hip >= 0.9 and r2_ref >= 0.9, (r2_hip, r2_ref)
    assert not np.any((np.asarray(res[0]) != 0) & (w == 0))                 # no false positives
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(res[i], ref[i]) < 1e-6, nm


def test_invariants_full_size():
    """Size-independent properties at BASELINE configs[2] (N=1000, K=2000, 50 iterations) -- SURVEY 8c list."""
    from circuitmap_amd.optimise import caviar
    sim, pri = _sim_case(1000, 2000, 2000)
    stim = sim["stim_matrix"]
    res = caviar(sim["psc"], stim, *prior_args(pri), iters=50, seed=0, save_histories=False)
    mu, beta, lam, shape, rate, phi, phi_cov, z = res[:8]
    assert np.all((lam >= 0) & (lam <= 1))
    assert np.all(lam[stim == 0] == 0)                                   # supp(lam) within supp(I)
    assert shape == pri["shape"] + 2000 / 2                              # exact
    assert np.all(phi > 0)
    dead = ~np.any(lam != 0, axis=1)
    assert np.all(z[np.any(lam >= 0.1, axis=0)] == 0)
    assert np.all(z >= 0)
    # (the R^2 >= 0.9 accuracy anchor is tested at the shape it is quoted on: test_accuracy_anchor_notebook_shape; this shape has 20
    # trials per cell, where the oracle and the HIP path alike reach 0.77 -- held to the oracle in test_caviar_vs_oracle_bench_sizes)
    assert dead.sum() > 0.5 * len(dead)                                  # most cells are disconnected and pruned


def test_stress_size_vs_oracle_and_invariants():
    """BASELINE configs[4] (N=5000, K=10000: 157-tile Cholesky with the panel-solve + bulk trailing-update kernels, the tiled
    inverse at scale, two-round shuffle, 8-bit chain tickets).  (1) The first 3 iterations + false-negative scan + final update_phi
    against the running-residual oracle on identical inputs (~100 s of CPU): every large-matrix kernel meets an oracle, not only
    invariants.  (2) The full 50 iterations: size-independent properties, plus: a row whose lam is identically 0 has exactly the
    prior mean and the prior VARIANCE (quirk: `beta` returns diag(Sigma)), and a second run of the same problem object reproduces
    the first bit for bit."""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem
    from oracle import caviar_oracle as co
    sim, pri = _sim_case(5000, 10000, 2005, lite=True)
    stim = sim["stim_matrix"]
    ctx = _lib.default_context(0)
    prob = Problem(ctx, sim["psc"], stim)
    args = prior_args(pri)
    prob.run(*args, iters=3, seed=0)
    res3 = prob.fetch()
    ref3 = co.caviar(sim["psc"], stim, *args, literal=False, iters=3, seed=0)
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(res3[i], ref3[i]) < 1e-6, nm
    del ref3, res3
    prob.run(*args, iters=50, seed=0)
    res = prob.fetch()
    mu, beta, lam, shape, rate, phi, phi_cov, z = res[:8]
    assert np.all((lam >= 0) & (lam <= 1))
    assert np.all(lam[stim == 0] == 0)
    assert shape == pri["shape"] + 10000 / 2
    assert np.all(phi > 0) and np.all(np.isfinite(phi_cov))
    assert np.all(z[np.any(lam >= 0.1, axis=0)] == 0) and np.all(z >= 0)
    dead = ~np.any(lam != 0, axis=1)
    assert dead.sum() > 0.5 * len(dead)
    assert np.all(mu[dead] == pri["mu"][dead]) and np.all(beta[dead] == pri["beta"][dead] ** 2)
    assert np.corrcoef(sim["weights"], mu)[0, 1] ** 2 > 0.6               # 20 trials per cell (anchor: test_accuracy_anchor_notebook_shape)
    prob.run(*args, iters=50, seed=0)
    res2 = prob.fetch()
    for a, b in zip(res[:8], res2[:8]):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    prob.close()


def test_edge_cases(ctx):
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise import caviar
    from oracle import caviar_oracle as co
    rng = np.random.default_rng(3)
    # (a) N = 2 (minimum), integer stimulus matrix, odd T, row never targeted
    stim = np.zeros((2, 30), dtype=int)
    stim[0, ::2] = 50
    psc = rng.normal(0, 0.2, (30, 7))
    pri = default_priors(2)
    res = caviar(psc, stim, *prior_args(pri), iters=4, num_mc_samples=3)
    ref = co.caviar(psc, stim, *prior_args(pri), iters=4, num_mc_samples=3, literal=True)
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(res[i], ref[i]) < 1e-9, nm
    # (b) all trials masked out (psc == 0): lam stays 0, mu = prior, no NaN
    res = caviar(np.zeros((30, 7)), stim, *prior_args(pri), iters=3)
    assert np.all(res[2] == 0) and np.all(res[0] == 0) and np.all(np.isfinite(res[5]))
    # (c) iters = 0 returns the priors and lam0 (caviar.py:32-34)
    res = caviar(psc, stim, *prior_args(pri), iters=0, fn_scan=False)
    assert np.array_equal(res[0], pri["mu"]) and np.array_equal(res[1], pri["beta"])
    # (d) too many distinct powers -> error code, not a crash
    many = rng.permutation(100 * 3).reshape(3, 100).astype(float)
    with pytest.raises(_lib.CaviarHipError):
        caviar(rng.normal(0, 1, (100, 5)), many, *prior_args(default_priors(3)), iters=1)
    # (e) unknown keyword -> TypeError like the reference's **fit_options
    with pytest.raises(TypeError):
        caviar(psc, stim, *prior_args(pri), minimax_spk_prob=0.3)


def test_ticket_timeout_is_reported_not_hung():
    """The multi-wave chain waits for per-trial tickets with a BOUNDED spin; a ticket that never comes (injected through
    CAVIAR_HIP_TEST_BAD_TICKET in a child process: the flag is read once per process) must end the run with an error, not hang the GPU
    and not return wrong results silently."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch\n"
        "from circuitmap_amd import _lib\n"
        "from circuitmap_amd.optimise import caviar\n"
        "from circuitmap_amd.simulation import simulate\n"
        "from helpers import default_priors, prior_args\n"
        "sim = simulate(N=40, trials=120, seed=3)\n"
        "try:\n"
        "    caviar(sim['psc'], sim['stim_matrix'], *prior_args(default_priors(40)), iters=3)\n"
        "    print('NO ERROR')\n"
        "except _lib.CaviarHipError as e:\n"
        "    print('ERROR:', e)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CAVIAR_HIP_TEST_BAD_TICKET="1", CAVIAR_HIP_FORCE_PAR_SWEEP="1")   # (at this small shape the dense chain would take the single-consumer kernel)
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    out = p.stdout.decode()
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert "ERROR:" in out and "ticket" in out, out


def test_model_interface():
    import circuitmap_amd as cm
    sim = cm.simulate(N=50, trials=200, seed=5)
    model = cm.Model(50)
    model.fit(sim["psc"], sim["stim_matrix"], method="caviar", fit_options={"iters": 5, "save_histories": True})
    assert set(model.state) >= {"mu", "beta", "shape", "rate", "phi", "phi_cov", "lam", "z", "receptive_fields", "alpha"}
    assert model.state["lam"].shape == (50, 200) and model.trial_count == 200 and model.time > 0
    assert model.history["lam"].shape == (5, 50, 200) and model.history["shape"].shape == (5, 200)
    with pytest.raises(Exception):
        model.fit(sim["psc"], sim["stim_matrix"], method="nope")


def _edge_problem(seed, N, K, T, H, powers, dense_rows=False, no_zero=False):
    rng = np.random.default_rng(seed)
    stim = np.zeros((N, K))
    for k in range(K):
        tars = rng.choice(N, H, replace=False)
        stim[tars, k] = powers[rng.integers(0, len(powers))]
    if no_zero:
        stim[stim == 0] = min(powers) / 2.0            # every entry targeted: unique(I)[1:] then drops the smallest power (quirk 8)
    w = np.zeros(N)
    w[: max(1, N // 4)] = rng.uniform(10, 30, max(1, N // 4))
    spk = (rng.random((N, K)) < 1 / (1 + np.exp(-(0.2 * stim - 9.0)))) * (stim > 0)
    t = np.arange(T)
    kern = np.exp(-t / 20.0) - np.exp(-t / 5.0)
    kern /= np.trapezoid(kern)
    psc = (w @ spk)[:, None] * kern[None, :] + rng.normal(0, 5e-3, (K, T))
    return psc, stim


@pytest.mark.parametrize("case", ["long_rows", "many_powers", "no_zero_entries", "single_target", "long_rows_many_powers"])
def test_rare_paths_vs_literal_oracle(case):
    """Paths the synthetic benchmark shapes never take: rows with more than 64 targeted trials (two-pass row step, tiled
    precision build with long rows), more than 8 distinct powers (literal pool-adjacent-violators walk on the device), a
    stimulus matrix without zeros (the reference then drops the smallest power from `powers`), one target per trial."""
    from circuitmap_amd.optimise import caviar
    from oracle import caviar_oracle as co
    cfg = {
        "long_rows": dict(N=12, K=240, T=50, H=6, powers=[40.0, 55.0, 70.0]),
        "many_powers": dict(N=16, K=200, T=50, H=3, powers=[30.0 + 4 * i for i in range(11)]),
        "no_zero_entries": dict(N=5, K=60, T=50, H=2, powers=[40.0, 55.0, 70.0], no_zero=True),
        "single_target": dict(N=20, K=160, T=50, H=1, powers=[45.0, 65.0]),
        "long_rows_many_powers": dict(N=10, K=300, T=40, H=5, powers=[30.0 + 5 * i for i in range(9)]),
    }[case]
    psc, stim = _edge_problem(11, **cfg)
    pri = default_priors(stim.shape[0])
    opts = dict(iters=8, num_mc_samples=9, seed=5, minimum_spike_count=2, msrmp=0.15)
    res = caviar(psc, stim, *prior_args(pri), **opts)
    ref = co.caviar(psc, stim, *prior_args(pri), literal=True, **opts)
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(res[i], ref[i]) < 1e-7, (case, nm)


@pytest.mark.parametrize("per_launch,streams", [(16, 1), (2, 3)])
def test_fit_batch_single_gpu_equals_direct(per_launch, streams):
    """batch.fit_batch without a process group == a loop of caviar() calls, bit for bit: as one lock-step batch, and as several
    batches in flight on several contexts (streams, one host thread each)."""
    from circuitmap_amd import batch
    from circuitmap_amd.optimise import caviar
    from circuitmap_amd.simulation import simulate
    circuits = []
    for c in range(5):
        sim = simulate(N=40, trials=120, seed=50 + c)
        circuits.append((sim["psc"], sim["stim_matrix"]))
    pri = default_priors(40)
    opts = dict(iters=6, seed=1)
    got = batch.fit_batch(circuits, priors=pri, fit_options=opts, circuits_per_launch=per_launch, streams=streams)
    assert len(got) == 5
    for (psc, stim), g in zip(circuits, got):
        res = caviar(psc, stim, *prior_args(pri), **opts)
        for i, nm in enumerate(OUT_NAMES):
            if nm != "lam":
                assert np.array_equal(np.asarray(res[i], dtype=float), np.asarray(g[nm], dtype=float), equal_nan=True), nm


def _assert_same_fit(a, b, what=""):
    for i, nm in enumerate(OUT_NAMES):
        np.testing.assert_array_equal(np.asarray(a[i], dtype=float), np.asarray(b[i], dtype=float), err_msg="%s %s" % (what, nm))


def test_run_batch_bit_identical_to_serial(ctx):
    """cm_caviar_run_batch: circuits of two different shapes in one call (-> two lock-step groups), false-negative scan on, every
    circuit bit-identical to its own cm_caviar_run; a second batch on the same problem objects reproduces the first; histories."""
    from circuitmap_amd.optimise.caviar import Problem, run_batch
    shapes = [(40, 120), (120, 400), (40, 120), (120, 400), (40, 120), (120, 400), (120, 400)]
    sims = [_sim_case(N, K, 700 + i)[0] for i, (N, K) in enumerate(shapes)]
    pris = [default_priors(N) for N, _ in shapes]
    pris[1]["beta"] = 2e1 * np.ones(120)                                  # per-circuit priors are honoured
    opts = dict(iters=14, seed=2)
    serial = []
    for sim, pri in zip(sims, pris):
        q = Problem(ctx, sim["psc"], sim["stim_matrix"])
        q.run(*prior_args(pri), **opts)
        serial.append(q.fetch())
        q.close()
    probs = [Problem(ctx, sim["psc"], sim["stim_matrix"]) for sim in sims]
    tms = run_batch(probs, pris, **opts)
    assert sorted(t.batch for t in tms) == [3, 3, 3, 4, 4, 4, 4]
    for q, ref in zip(probs, serial):
        _assert_same_fit(q.fetch(), ref, "batch")
    run_batch(probs[::-1], pris[::-1], **opts)                            # again, other order: state fully re-initialised
    for q, ref in zip(probs, serial):
        _assert_same_fit(q.fetch(), ref, "second batch")
    # histories through the batch path
    hopts = dict(iters=5, seed=1, save_histories=True)
    run_batch(probs[:3:2], [pris[0], pris[2]], **hopts)
    for q, sim, pri in [(probs[0], sims[0], pris[0]), (probs[2], sims[2], pris[2])]:
        q1 = Problem(ctx, sim["psc"], sim["stim_matrix"])
        q1.run(*prior_args(pri), **hopts)
        a, b = q.fetch(), q1.fetch()
        for i in range(8):
            np.testing.assert_array_equal(a[i], b[i])
            np.testing.assert_array_equal(a[9 + i], b[9 + i])
        q1.close()
    for q in probs:
        q.close()


@pytest.mark.parametrize("case", ["wide_trials", "long_rows", "odd_counts"])
def test_run_batch_rare_shapes_bit_identical(ctx, case):
    """Lock-step batches on shapes that take the batch kernels' other paths -- trials with more than 16 targeted neurons (the
    two-columns-per-segment column sums fall back to one column after the other), rows of more than 32 / 64 trials (rows one per
    wavefront; the general pre-sweep instantiation), odd N and K (unpaired last row / column) -- every circuit bit-identical to its
    own single fit, and the single fit within 1e-7 of the literal oracle."""
    from circuitmap_amd.optimise.caviar import Problem, run_batch
    from circuitmap_amd.optimise import caviar
    from oracle import caviar_oracle as co
    cfg = {"wide_trials": dict(N=60, K=90, T=40, H=20, powers=[40.0, 55.0, 70.0]),
           "long_rows": dict(N=30, K=400, T=40, H=6, powers=[40.0, 55.0, 70.0]),
           "odd_counts": dict(N=41, K=121, T=40, H=5, powers=[45.0, 65.0])}[case]
    opts = dict(iters=8, num_mc_samples=9, seed=5, minimum_spike_count=2, msrmp=0.15)
    probs_in = [_edge_problem(21 + i, **cfg) for i in range(3)]
    pri = default_priors(cfg["N"])
    singles = [caviar(psc, stim, *prior_args(pri), **opts) for psc, stim in probs_in]
    ref = co.caviar(probs_in[0][0], probs_in[0][1], *prior_args(pri), literal=True, **opts)
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(singles[0][i], ref[i]) < 1e-7, (case, nm)
    probs = [Problem(ctx, psc, stim) for psc, stim in probs_in]
    run_batch(probs, pri, **opts)
    for q, single in zip(probs, singles):
        got = q.fetch()
        for i, nm in enumerate(OUT_NAMES):
            np.testing.assert_array_equal(np.asarray(got[i], dtype=float), np.asarray(single[i], dtype=float), err_msg="%s %s" % (case, nm))
        q.close()


def test_run_batch_listed_rows_overflow(ctx, monkeypatch):
    """The steady-state per-neuron role of a batch covers the rows with lam != 0 with a grid sized from the previous active set; rows
    beyond it are taken by k_iter_overflow.  CAVIAR_HIP_TEST_LISTED_ROWS=4 makes the grid four rows, so nearly every listed row goes
    through the overflow kernel: results must still equal the single fits bit for bit."""
    from circuitmap_amd.optimise.caviar import Problem, run_batch
    from circuitmap_amd.optimise import caviar
    sims = [_sim_case(120, 400, 900 + i)[0] for i in range(3)]
    pri = default_priors(120)
    opts = dict(iters=14, seed=3)
    singles = [caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), **opts) for sim in sims]
    monkeypatch.setenv("CAVIAR_HIP_TEST_LISTED_ROWS", "4")
    probs = [Problem(ctx, sim["psc"], sim["stim_matrix"]) for sim in sims]
    run_batch(probs, pri, **opts)
    monkeypatch.delenv("CAVIAR_HIP_TEST_LISTED_ROWS")
    for q, single in zip(probs, singles):
        _assert_same_fit(q.fetch(), single, "listed-rows overflow")
        q.close()


def test_run_batch_explicit_streams_and_argument_checks(ctx):
    """A batch fed explicit RNG streams (shared by its circuits) equals the seeded batch bit for bit; argument errors come back as
    error codes (problems of another context, a problem listed twice, NULL priors), not as faults."""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem, run_batch
    from oracle import threefry
    sims = [_sim_case(60, 150, 950 + i)[0] for i in range(3)]
    pri = default_priors(60)
    opts = dict(iters=7, seed=5, num_mc_samples=20)
    orders, u = threefry.make_explicit_streams(5, 7, 60, 20)
    probs = [Problem(ctx, s_["psc"], s_["stim_matrix"]) for s_ in sims]
    run_batch(probs, pri, **opts)
    seeded = [q.fetch() for q in probs]
    run_batch(probs, pri, update_orders=orders, uniforms=u, **opts)
    for q, ref in zip(probs, seeded):
        _assert_same_fit(q.fetch(), ref, "explicit streams")
    other = _lib.Context(0)
    foreign = Problem(other, sims[0]["psc"], sims[0]["stim_matrix"])
    with pytest.raises(_lib.CaviarHipError):
        run_batch([probs[0], foreign], pri, **opts)                      # mixed contexts are refused in Python ...
    handles = (C.c_void_p * 2)(probs[0].handle, foreign.handle)
    o = _lib.CmOptions(); ctx.lib.cm_options_default(C.byref(o))
    keep = []
    from circuitmap_amd.optimise.caviar import _priors_struct
    prs = (_lib.CmPriors * 2)(*[_priors_struct(60, *prior_args(pri), keep) for _ in range(2)])
    assert ctx.lib.cm_caviar_run_batch(ctx.handle, 2, handles, prs, C.byref(o), None) == -1     # ... and by the C ABI (CM_ERR_INVALID)
    handles = (C.c_void_p * 2)(probs[0].handle, probs[0].handle)
    assert ctx.lib.cm_caviar_run_batch(ctx.handle, 2, handles, prs, C.byref(o), None) == -1     # the same problem twice
    assert b"twice" in ctx.lib.cm_last_error()
    bad = (_lib.CmPriors * 1)(_lib.CmPriors())
    assert ctx.lib.cm_caviar_run_batch(ctx.handle, 1, (C.c_void_p * 1)(probs[0].handle), bad, C.byref(o), None) == -1   # NULL prior arrays
    # cm_caviar_fetch_batch refuses a problem of another context (its results are ordered on that context's stream: ADVICE r3)
    foreign.run(*prior_args(pri), **opts)
    rs = (_lib.CmResult * 2)()
    bufs = [np.empty(60), np.empty(60)]
    for r_, b_ in zip(rs, bufs):
        r_.mu = _lib.dptr(b_)
    assert ctx.lib.cm_caviar_fetch_batch(ctx.handle, 2, (C.c_void_p * 2)(probs[0].handle, foreign.handle), rs) == -1
    assert b"another context" in ctx.lib.cm_last_error()
    foreign.close(); other.close()
    for q in probs:
        q.close()


def test_fit_batch_baseline_config3():
    """BASELINE configs[3]: the batch of 8 independent N=1000 / K=2000 circuits (seeds 2000..2007), 50 iterations + fn_scan, here
    all on ONE GPU as one lock-step batch through batch.fit_batch; every circuit bit-identical to its direct caviar() call."""
    from circuitmap_amd import batch
    from circuitmap_amd.optimise import caviar
    circuits = []
    for c in range(8):
        sim, pri = _sim_case(1000, 2000, 2000 + c)
        circuits.append((sim["psc"], sim["stim_matrix"]))
    opts = dict(iters=50, seed=0)
    got = batch.fit_batch(circuits, priors=pri, fit_options=opts)
    assert len(got) == 8
    for (psc, stim), g in zip(circuits, got):
        res = caviar(psc, stim, *prior_args(pri), **opts)
        for i, nm in enumerate(OUT_NAMES):
            if nm != "lam":
                np.testing.assert_array_equal(np.asarray(res[i], dtype=float), np.asarray(g[nm], dtype=float), err_msg=nm)
        assert np.count_nonzero(res[0]) > 10                              # a real fit: connections were found
    # compact=True: float32 traces + stimulus triples over PCIe -- bitwise the dense float64 path on float32-representable traces
    c32 = [(np.asarray(psc, dtype=np.float32).astype(np.float64), stim) for psc, stim in circuits[:2]]
    dense = batch.fit_batch(c32, priors=pri, fit_options=opts)
    comp = batch.fit_batch(c32, priors=pri, fit_options=opts, compact=True)
    for d_, c_ in zip(dense, comp):
        for nm in ("mu", "beta", "phi", "phi_cov", "z", "shape", "rate"):
            np.testing.assert_array_equal(np.asarray(d_[nm], dtype=float), np.asarray(c_[nm], dtype=float), err_msg="compact " + nm)


def test_fit_batch_c_entry_point(ctx):
    """cm_caviar_fit_batch (host arrays in, host results out, device list) and cm_caviar_fit == caviar()."""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise import caviar
    from circuitmap_amd.optimise.caviar import make_options, _priors_struct
    lib = ctx.lib
    sims = [_sim_case(60, 150, 900 + i)[0] for i in range(3)]
    pri = default_priors(60)
    opts = dict(iters=9, seed=4)
    o = make_options(lib, **opts)
    keep = []
    n = len(sims)
    psc = [_lib.f64(s_["psc"]) for s_ in sims]
    stim = [_lib.f64(s_["stim_matrix"]) for s_ in sims]
    prs = (_lib.CmPriors * n)(*[_priors_struct(60, *prior_args(pri), keep) for _ in range(n)])
    outs, res = [], (_lib.CmResult * n)()
    for c in range(n):
        out = dict(mu=np.empty(60), beta=np.empty(60), lam=np.empty((60, 150)), phi=np.empty((60, 2)), phi_cov=np.empty((60, 2, 2)), z=np.empty(150))
        for k, v in out.items():
            setattr(res[c], k, _lib.dptr(v))
        outs.append(out)
    i32 = lambda v: (C.c_int32 * n)(*v)
    dpp = lambda arrs: (C.POINTER(C.c_double) * n)(*[_lib.dptr(a) for a in arrs])
    _lib.check(lib.cm_caviar_fit_batch(n, i32([60] * n), i32([150] * n), i32([900] * n), dpp(psc), dpp(stim), prs, C.byref(o), None, 0, res, None))
    for c in range(n):
        ref = caviar(sims[c]["psc"], sims[c]["stim_matrix"], *prior_args(pri), **opts)
        for i, nm in enumerate(["mu", "beta", "lam"]):
            np.testing.assert_array_equal(outs[c][nm], ref[i], err_msg=nm)
        assert res[c].shape == ref[3] and res[c].rate == ref[4]
    # cm_caviar_fit: create + run + fetch + destroy in one call
    one = dict(mu=np.empty(60), lam=np.empty((60, 150)), z=np.empty(150))
    r1 = _lib.CmResult()
    for k, v in one.items():
        setattr(r1, k, _lib.dptr(v))
    tm = _lib.CmTimings()
    _lib.check(lib.cm_caviar_fit(ctx.handle, 60, 150, 900, _lib.dptr(psc[0]), _lib.dptr(stim[0]), C.byref(prs[0]), C.byref(o), C.byref(r1), C.byref(tm)))
    ref = caviar(sims[0]["psc"], sims[0]["stim_matrix"], *prior_args(pri), **opts)
    np.testing.assert_array_equal(one["mu"], ref[0]); np.testing.assert_array_equal(one["lam"], ref[2]); np.testing.assert_array_equal(one["z"], ref[7])
    assert tm.iters == 9 and tm.batch == 1 and tm.total_ms > 0


def test_phase_hooks_vs_golden(ctx):
    """cm_k_sigma / cm_k_spont / cm_k_lam_sweep against the per-phase vectors recorded from the reference's source (units.npz):
    update_sigma (caviar.py:238-244), the soft threshold at outer iterations 0, 5, 19, 20 (quirk 4: frozen from 20 on), one
    update_lam sweep ungated and gated incl. the carried PRNG key (caviar.py:190-231)."""
    from circuitmap_amd import _lib
    lib = ctx.lib
    g = np.load(GOLDEN + "/units.npz")
    stim, lam, y = g["stim"], g["lam"], g["y"]
    mask = np.ascontiguousarray(g["mask"].astype(np.uint8))
    N, K = stim.shape
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    mu, beta = np.ascontiguousarray(g["bum_mu"]), np.ascontiguousarray(g["bum_beta"])
    sh, rt = C.c_double(), C.c_double()
    _lib.check(lib.cm_k_sigma(ctx.handle, N, K, _lib.dptr(stim), _lib.dptr(lam), _lib.dptr(y), _lib.dptr(mu), _lib.dptr(beta), 1.0, 0.1,
                              C.byref(sh), C.byref(rt)))
    assert sh.value == g["sig_shape"] and abs(rt.value - g["sig_rate"]) <= 1e-12 * abs(g["sig_rate"])
    for it in (0, 5, 19, 20):
        z = np.zeros(K) + 0.123
        sr = C.c_double()
        _lib.check(lib.cm_k_spont(ctx.handle, N, K, _lib.dptr(stim), _lib.dptr(lam), _lib.dptr(y), u8(mask), _lib.dptr(mu), it, 20, 5.0, 0.75, 0.05,
                                  0.1, _lib.dptr(z), C.byref(sr)))
        assert rel_err(z, g["z_it%d" % it]) < 1e-12, it
        assert np.array_equal(z != 0, g["z_it%d" % it] != 0) and sr.value == np.mean(g["z_it%d" % it] != 0)
    pri = default_priors(N)
    for tag, it in (("ungated", 0), ("gated", 5)):
        lam_io = lam.copy()
        key = np.ascontiguousarray(g["ul_key_in"].astype(np.uint32))
        _lib.check(lib.cm_k_lam_sweep(ctx.handle, N, K, _lib.dptr(stim), _lib.dptr(lam_io), _lib.dptr(np.ascontiguousarray(g["ul_y"])), u8(mask),
                                      _lib.dptr(np.ascontiguousarray(g["ul_mu"])), _lib.dptr(np.ascontiguousarray(g["ul_beta"])), 3.0, 0.7,
                                      _lib.dptr(np.ascontiguousarray(g["phi"])), _lib.dptr(np.ascontiguousarray(g["phi_cov"])),
                                      key.ctypes.data_as(C.POINTER(C.c_uint32)), 7, 2.0, 0.35, it, 1))
        assert np.array_equal(key, g["ul_key_" + tag])
        assert rel_err(lam_io, g["ul_lam_" + tag]) < 1e-10, tag
        assert np.array_equal(lam_io != 0, g["ul_lam_" + tag] != 0)
    assert np.any(g["ul_lam_gated"].sum(axis=1) == 0) and np.any(g["ul_lam_gated"] != g["ul_lam_ungated"])   # the gate did close rows


@pytest.mark.parametrize("name", ["caviar_cfg2", "caviar_cfg3"])
def test_caviar_vs_golden_bench_shapes(name):
    """BASELINE configs[1] (N=300, K=1000, T=900; first 12 iterations) and configs[2] -- the HEADLINE shape N=1000, K=2000, T=900; first 8
    iterations: three ungated full-size tiled solves, the gating transition, the collapsing active set -- against the vectors recorded from the
    reference's own source text (tests/golden/make_golden.py: case_cfg2, case_cfg3; caviar.py:79-98): iterations + false-negative scan + final
    update_phi, all per-iteration histories (lam stored on the support of the stimulus matrix)."""
    from circuitmap_amd.optimise import caviar
    g, psc, stim, pri, opts = load_case(name)
    res = caviar(psc, stim, *prior_args(pri), **opts)
    sup = stim > 0
    for i, nm in enumerate(OUT_NAMES):
        got, hist = np.asarray(res[i], dtype=float), np.asarray(res[9 + i], dtype=float)
        if nm == "lam":
            assert not np.any(got[~sup]) and not np.any(hist[:, ~sup])
            got, hist = got[sup], hist[:, sup]
        if nm in ("shape", "rate"):
            assert np.all(hist == hist[:, :1])
            hist = hist[:, 0]
        assert rel_err(got, g["out_" + nm]) < TOL_FINAL * 100, nm             # 12 iterations: still inside the amplified early phase
        assert rel_err(hist, g["out_hist_" + nm]) < TOL_HIST, "hist " + nm


# ---------------------------------------------------------------------------------------------------------------------------------
# The launch configuration bench.py times: lock-step batches with SEVERAL circuits per XCD (B = 16 ... 256), the listed-rows grid at
# that size, the library's default scheduling (two interleaved half-batches), and the same circuits as 4 x 32 on four contexts.  (Unit of scale-out this replaces: one process
# per circuit, reference scripts/generate_loho_cv_slurm_scripts.py:8-36.)
# ---------------------------------------------------------------------------------------------------------------------------------
def _device_circuits(cx, N, K, B, seed0, T=900, host_copy_of=()):
    """B circuits generated on the device (include/sim_hip.h) as Problems of context `cx`; host copies (psc, stim) of the listed ones"""
    from circuitmap_amd.optimise.caviar import Problem
    from circuitmap_amd.simulation import simulate_device
    probs, host = [], {}
    for c in range(B):
        d = simulate_device(N=N, T=T, H=10, trials=K, seed=seed0 + c, keep_on_device=True, ctx=cx)
        if c in host_copy_of:
            host[c] = (cx.download(d["psc_ptr"], np.empty((K, T))), cx.download(d["stim_ptr"], np.empty((N, K))))
        probs.append(Problem(cx, None, None, device_ptrs=(N, K, T, d["psc_ptr"], d["stim_ptr"])))
        cx.free(d["psc_ptr"]); cx.free(d["stim_ptr"])
    return probs, host


@pytest.mark.parametrize("N,K,B,iters", [(200, 400, 16, 14), (200, 400, 33, 14), (200, 400, 128, 14), (200, 400, 520, 14), (1000, 2000, 16, 50), (1000, 2000, 256, 50),
                                         (1000, 2000, 512, 50), (200, 3600, 16, 14), (5000, 10000, 8, 3)])
def test_run_batch_many_per_xcd(ctx, N, K, B, iters):
    """cm_caviar_run_batch with 2 and 16 circuits per XCD (cm_xcd_remap mode 3 with per > 1, which needs gridDim.y % 8 == 0): every
    circuit bit-identical to its own cm_caviar_run; circuit 0 within 1e-6 of the running-residual oracle (the small shape; the
    N = 1000 oracle trajectory is test_caviar_vs_oracle_bench_sizes).  N = 200 / K = 400 starts with a 7-tile-row tiled solve; N = 200 / K = 3600
    is the reference's published workload (examples/demo_compressive_connectivity_mapping.ipynb:172-175: ~180 trials per cell, every chain row is
    three 64-entry record segments in the multi-wave chain kernel); N = 5000 / K = 10000 is BASELINE configs[4] as a batch (bench.py --workload cfg5
    --circuits-per-gpu 16 / 64): 157-tile-row matrices through the grouped Cholesky family against the single fits' pair family, the false-negative
    scan on the global arrays (eight circuits) against the single fits' host scan;
    (1000, 2000, 512) is bench.py's default step ((1000, 2000, 256) was, until round 4): timed as one lock-step batch, and as the two interleaved
    half-batches that are the library's default scheduling (cm_options.interleave) -- both forms are checked; B = 520 is 65 circuits per XCD."""
    from circuitmap_amd.optimise.caviar import run_batch
    from oracle import caviar_oracle as co
    pri = default_priors(N)
    opts = dict(iters=iters, seed=0)
    probs, host = _device_circuits(ctx, N, K, B, 4000, host_copy_of=(0,) if N <= 200 else ())
    tms = run_batch(probs, pri, interleave=False, **opts)                 # one lock-step batch of B
    assert all(t.batch == B for t in tms)
    batch = [q.fetch() for q in probs]
    tms = run_batch(probs, pri, **opts)                                   # the default: two half-batches at a time on two streams (B >= 32)
    halves = [B] * B if B < 32 else [(B + 1) // 2] * ((B + 1) // 2) + [B // 2] * (B // 2)
    assert [t.batch for t in tms] == halves
    inter = [q.fetch() for q in probs]
    for c, q in enumerate(probs):
        q.run(*prior_args(pri), **opts)
        single = q.fetch()
        _assert_same_fit(batch[c], single, "B=%d circuit %d" % (B, c))
        _assert_same_fit(inter[c], single, "B=%d (two interleaved halves) circuit %d" % (B, c))
    if host:
        ref = co.caviar(host[0][0], host[0][1], *prior_args(pri), literal=False, **opts)
        for i, nm in enumerate(OUT_NAMES):
            assert rel_err(batch[0][i], ref[i]) < 1e-6, nm
    assert np.count_nonzero(batch[0][0]) > 0
    for q in probs:
        q.close()


def test_create_batch_and_fetch_batch(ctx):
    """cm_problem_create_batch (prologue kernels once for a group, circuits of different shapes, host arrays and device pointers) gives the
    encoding of cm_problem_create -- the fits are bitwise the same -- and cm_caviar_fetch_batch the numbers of cm_caviar_fetch; a failing
    circuit (too many distinct stimulus values) fails the whole call with the reference's error and leaves nothing behind."""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem, fetch_batch, run_batch
    shapes = [(40, 120), (120, 400), (40, 120), (200, 400), (120, 400)]
    sims = [_sim_case(N, K, 900 + i)[0] for i, (N, K) in enumerate(shapes)]
    pris = [default_priors(N) for N, _ in shapes]
    opts = dict(iters=10, seed=3)
    one_by_one = [Problem(ctx, sim["psc"], sim["stim_matrix"]) for sim in sims]
    grouped = Problem.create_batch(ctx, [(sim["psc"], sim["stim_matrix"]) for sim in sims])
    assert [q.nnz for q in grouped] == [q.nnz for q in one_by_one]
    run_batch(one_by_one, pris, **opts)
    run_batch(grouped, pris, **opts)
    packed = fetch_batch(grouped)
    for a, b, pk in zip(one_by_one, grouped, packed):
        ra, rb = a.fetch(), b.fetch()
        _assert_same_fit(ra, rb, "create_batch")
        for i, nm in enumerate(OUT_NAMES):
            if nm != "lam":
                np.testing.assert_array_equal(np.asarray(rb[i], dtype=float), np.asarray(pk[i], dtype=float), err_msg="fetch_batch " + nm)
    for q in one_by_one + grouped:
        q.close()
    # device pointers (the path of fit_batch(scatter=True) and of bench.py's end-to-end leg)
    N, K = shapes[1]
    dev = []
    for sim in sims[1::3]:
        psc, stim = np.ascontiguousarray(sim["psc"], dtype=float), np.ascontiguousarray(sim["stim_matrix"], dtype=float)
        pp, sp = ctx.alloc(psc.nbytes), ctx.alloc(stim.nbytes)
        ctx.upload(pp, psc); ctx.upload(sp, stim)
        dev.append(("device", N, K, psc.shape[1], pp, sp))
    dprobs = Problem.create_batch(ctx, dev)
    for d in dev:
        ctx.free(d[4]); ctx.free(d[5])                                   # the problems keep their own encoding
    hprobs = [Problem(ctx, sim["psc"], sim["stim_matrix"]) for sim in sims[1::3]]
    run_batch(dprobs, pris[1], **opts)
    run_batch(hprobs, pris[1], **opts)
    for a, b in zip(hprobs, dprobs):
        _assert_same_fit(a.fetch(), b.fetch(), "device create_batch")
    for q in dprobs + hprobs:
        q.close()
    bad = sims[0]["stim_matrix"].astype(float).copy()
    bad[0, :100] = np.arange(1, 101) * 0.5                               # 100 distinct powers > CM_MAX_VALS
    with pytest.raises(_lib.CaviarHipError) as ei:
        Problem.create_batch(ctx, [(sims[1]["psc"], sims[1]["stim_matrix"]), (sims[0]["psc"], bad)])
    assert "distinct values" in str(ei.value)
    ctx.release_cache()                                                  # the pool hands everything back


def test_compact_inputs_build_the_same_circuits(ctx):
    """cm_problem_create_batch_ex / cm_caviar_fit_batch_ex: psc as float32, the stimulus matrix as (trial, neuron, power) triples or dense float32
    (what the reference's callers hold before `.astype(float)`, scripts/run_downsampling_experiments.py:51) -> fits bitwise those of the dense
    float64 arrays of the same values; a triple outside the matrix is an error code, an all-zero stimulus matrix (no triples) a valid circuit."""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem, make_options, run_batch, _priors_struct
    lib = ctx.lib
    shapes = [(60, 150), (120, 400), (60, 150)]
    sims = [_sim_case(N, K, 990 + i)[0] for i, (N, K) in enumerate(shapes)]
    circuits = [(np.asarray(s_["psc"], dtype=np.float32).astype(np.float64), s_["stim_matrix"]) for s_ in sims]     # float32-representable traces
    pris = [default_priors(N) for N, _ in shapes]
    opts = dict(iters=8, seed=2)
    dense = Problem.create_batch(ctx, circuits)
    run_batch(dense, pris, **opts)
    want = [q.fetch() for q in dense]
    comp = Problem.create_batch(ctx, circuits, compact=True)
    assert [q.nnz for q in comp] == [q.nnz for q in dense]
    run_batch(comp, pris, **opts)
    for q, w in zip(comp, want):
        _assert_same_fit(q.fetch(), w, "compact inputs")
    # dense float32 stimulus matrix + float32 psc through the one-call entry point
    n = len(circuits)
    keep = []
    ins = (_lib.CmInputs * n)()
    for i, (psc, stim) in enumerate(circuits):
        p32, s32 = np.ascontiguousarray(psc, dtype=np.float32), np.ascontiguousarray(stim, dtype=np.float32)
        keep += [p32, s32]
        ins[i] = _lib.CmInputs(_lib.CM_DTYPE_F32, _lib.CM_STIM_DENSE_F32, p32.ctypes.data, s32.ctypes.data, 0)
    o = make_options(lib, **opts)
    prs = (_lib.CmPriors * n)(*[_priors_struct(N, *prior_args(pri), keep) for (N, _), pri in zip(shapes, pris)])
    res, outs = (_lib.CmResult * n)(), []
    for c, (N, K) in enumerate(shapes):
        ob = dict(mu=np.empty(N), beta=np.empty(N), lam=np.empty((N, K)), phi=np.empty((N, 2)), phi_cov=np.empty((N, 2, 2)), z=np.empty(K))
        for k, v in ob.items():
            setattr(res[c], k, _lib.dptr(v))
        outs.append(ob)
    i32 = lambda v: (C.c_int32 * n)(*v)
    _lib.check(lib.cm_caviar_fit_batch_ex(n, i32([s_[0] for s_ in shapes]), i32([s_[1] for s_ in shapes]), i32([900] * n), ins, prs, C.byref(o), None, 0, res, None))
    for c in range(n):
        for i, nm in enumerate(["mu", "beta", "lam"]):
            np.testing.assert_array_equal(outs[c][nm], want[c][i], err_msg=nm)
        assert res[c].shape == want[c][3] and res[c].rate == want[c][4]
    # a triple outside the matrix
    bad = np.array([(0, 0, 50.0), (150, 3, 50.0)], dtype=np.dtype([("trial", np.int32), ("neuron", np.int32), ("power", np.float32)]))
    p32 = np.ascontiguousarray(circuits[0][0], dtype=np.float32)
    one = (_lib.CmInputs * 1)(_lib.CmInputs(_lib.CM_DTYPE_F32, _lib.CM_STIM_TRIPLES, p32.ctypes.data, bad.ctypes.data, 2))
    h = (C.c_void_p * 1)()
    assert lib.cm_problem_create_batch_ex(ctx.handle, 1, (C.c_int32 * 1)(60), (C.c_int32 * 1)(150), (C.c_int32 * 1)(900), one, _lib.CM_MEM_HOST, h) == -1
    assert b"outside" in lib.cm_last_error()
    # no triples at all: a circuit nobody targets (lam stays 0, mu stays the prior)
    one = (_lib.CmInputs * 1)(_lib.CmInputs(_lib.CM_DTYPE_F32, _lib.CM_STIM_TRIPLES, p32.ctypes.data, None, 0))
    _lib.check(lib.cm_problem_create_batch_ex(ctx.handle, 1, (C.c_int32 * 1)(60), (C.c_int32 * 1)(150), (C.c_int32 * 1)(900), one, _lib.CM_MEM_HOST, h))
    lib.cm_problem_destroy(C.c_void_p(h[0]))
    for q in dense + comp:
        q.close()


def test_staged_upload_builds_the_same_circuits(ctx):
    """Host inputs reach the device through pinned slots filled by copy threads (4 MB pieces; the bytes of group g + 1 of 64 circuits travel while
    group g is encoded): 70 small circuits (two groups) and one whose arrays span several pieces are, bit for bit, the circuits built from the same
    arrays already in HBM; cm_caviar_fit_batch keeps its context between calls and cm_caviar_fit_batch_release drops it."""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem, make_options, run_batch, _priors_struct
    lib = ctx.lib
    shapes = [(40, 100)] * 69 + [(300, 1300)]                       # psc of the last one: 1300 x 900 x 8 B = 9.4 MB = three pieces
    sims = [_sim_case(N, K, 1200 + i)[0] for i, (N, K) in enumerate(shapes[:3])] + [_sim_case(300, 1300, 1290)[0]]
    pick = lambda i: sims[3] if i == 69 else sims[i % 3]
    rng = np.random.default_rng(5)
    circuits = []
    for i in range(70):
        s_ = pick(i)
        circuits.append((np.ascontiguousarray(s_["psc"] * (1.0 + 0.01 * (i // 3))), np.ascontiguousarray(s_["stim_matrix"], dtype=np.float64)))
    host = Problem.create_batch(ctx, circuits)
    ptrs = []
    for psc, stim in circuits:
        a, b = ctx.alloc(psc.nbytes), ctx.alloc(stim.nbytes)
        ctx.upload(a, psc); ctx.upload(b, stim)
        ptrs.append((a, b))
    ctx.synchronize()
    dev = Problem.create_batch(ctx, [("device", st.shape[0], st.shape[1], ps.shape[1], a, b) for (ps, st), (a, b) in zip(circuits, ptrs)])
    assert [q.nnz for q in host] == [q.nnz for q in dev]
    idx = [0, 1, 63, 64, 68, 69]
    opts = dict(iters=6, seed=4)
    for grp in (host, dev):
        run_batch([grp[i] for i in idx[:-1]], [default_priors(40)] * 5, **opts)
        run_batch([grp[69]], [default_priors(300)], **opts)
    for i in idx:
        _assert_same_fit(host[i].fetch(), dev[i].fetch(), "staged upload, circuit %d" % i)
    for q in host + dev:
        q.close()
    for a, b in ptrs:
        ctx.free(a); ctx.free(b)
    # the one-call entry point twice (second call: the kept context), then the release
    n = 3
    keep = []
    o = make_options(lib, **opts)
    prs = (_lib.CmPriors * n)(*[_priors_struct(40, *prior_args(default_priors(40)), keep) for _ in range(n)])
    i32 = lambda v: (C.c_int32 * n)(*v)
    dpp = lambda arrs: (C.POINTER(C.c_double) * n)(*[_lib.dptr(x) for x in arrs])
    mus = []
    for rep in range(2):
        res = (_lib.CmResult * n)()
        mu = [np.empty(40) for _ in range(n)]
        for c in range(n):
            res[c].mu = _lib.dptr(mu[c])
        _lib.check(lib.cm_caviar_fit_batch(n, i32([40] * n), i32([100] * n), i32([900] * n), dpp([c[0] for c in circuits[:n]]), dpp([c[1] for c in circuits[:n]]),
                                           prs, C.byref(o), None, 0, res, None))
        mus.append(mu)
    for c in range(n):
        np.testing.assert_array_equal(mus[0][c], mus[1][c])
    # from 2 x 64 circuits on the call is a two-stage pipeline (first half fitted by the helper thread on the context while the second is uploaded and
    # then fitted on the twin context): 130 circuits -> halves of 128 and 2; every mu bitwise that of the same circuit in an ordinary batch
    n = 130
    big = [circuits[i % 69] for i in range(n)]
    probs = Problem.create_batch(ctx, big)
    run_batch(probs, [default_priors(40)] * n, **opts)
    want = [q.fetch()[0] for q in probs]
    for q in probs:
        q.close()
    prs = (_lib.CmPriors * n)(*[_priors_struct(40, *prior_args(default_priors(40)), keep) for _ in range(n)])
    res = (_lib.CmResult * n)()
    mu = [np.empty(40) for _ in range(n)]
    zz = [np.empty(100) for _ in range(n)]
    for c in range(n):
        res[c].mu = _lib.dptr(mu[c]); res[c].z = _lib.dptr(zz[c])
    _lib.check(lib.cm_caviar_fit_batch(n, i32([40] * n), i32([100] * n), i32([900] * n), dpp([c[0] for c in big]), dpp([c[1] for c in big]),
                                       prs, C.byref(o), None, 0, res, None))
    for c in range(n):
        np.testing.assert_array_equal(mu[c], want[c], err_msg="pipelined fit_batch, circuit %d" % c)
    assert all(res[c].shape == res[0].shape for c in range(n))
    assert lib.cm_caviar_fit_batch_release() == 0
    assert lib.cm_caviar_fit_batch_release() == 0                   # nothing kept: still fine


def test_listed_column_sums_never_matter(ctx, monkeypatch):
    """The column sums R = lam^T mu / colany over the LISTED trials (k_stats lists the trials of the rows with lam != 0, the pre-sweep adds those of
    revived rows, everything else is zeroed) give bitwise the results of the walk over all columns: forced on for a small shape
    (CAVIAR_HIP_TEST_TRIAL_LISTS; by default the lists are used from K = 1536 on), off (CAVIAR_HIP_NO_TRIAL_LISTS is read once per process, so the
    reference here is the single fit of every circuit, which at this shape runs without lists), with a non-zero prior mean (lists must stay off:
    every row then has mu != 0) and against the oracle."""
    from circuitmap_amd.optimise.caviar import run_batch
    from oracle import caviar_oracle as co
    N, K, B = 200, 400, 8
    pri = default_priors(N)
    opts = dict(iters=20, seed=1)
    probs, host = _device_circuits(ctx, N, K, B, 5200, host_copy_of=(0,))
    singles = []
    for q in probs:
        q.run(*prior_args(pri), **opts)
        singles.append(q.fetch())
    monkeypatch.setenv("CAVIAR_HIP_TEST_TRIAL_LISTS", "1")
    run_batch(probs, pri, **opts)
    for c, q in enumerate(probs):
        _assert_same_fit(q.fetch(), singles[c], "listed trials, circuit %d" % c)
    probs[0].run(*prior_args(pri), **opts)                                # one circuit with lists
    _assert_same_fit(probs[0].fetch(), singles[0], "listed trials, single fit")
    ref = co.caviar(host[0][0], host[0][1], *prior_args(pri), literal=False, **opts)
    for i, nm in enumerate(OUT_NAMES):
        assert rel_err(singles[0][i], ref[i]) < 1e-6, nm
    pri2 = dict(pri, mu=0.5 * np.ones(N))                                 # prior mean != 0: all rows carry mu != 0, the lists must not be used
    run_batch(probs[:2], pri2, **opts)
    with_env = [q.fetch() for q in probs[:2]]
    monkeypatch.delenv("CAVIAR_HIP_TEST_TRIAL_LISTS")
    run_batch(probs[:2], pri2, **opts)
    for c in range(2):
        _assert_same_fit(probs[c].fetch(), with_env[c], "prior mean != 0, circuit %d" % c)
    for q in probs:
        q.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_run_batch_xcd_placement_never_matters(ctx, monkeypatch, mode):
    """CAVIAR_HIP_XCD_MODE = 0 (plain grid), 1 (interleaved), 2 (circuit-major) give bitwise the results of the default mode 3:
    the workgroup -> (tile, circuit) map is a speed heuristic only."""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import run_batch
    N, K, B = 200, 400, 16
    pri = default_priors(N)
    opts = dict(iters=8, seed=0)
    probs, _ = _device_circuits(ctx, N, K, B, 4000)
    run_batch(probs, pri, **opts)
    ref = [q.fetch() for q in probs]
    for q in probs:
        q.close()
    monkeypatch.setenv("CAVIAR_HIP_XCD_MODE", str(mode))
    other = _lib.Context(0)                                   # the mode is read when a context is created
    monkeypatch.delenv("CAVIAR_HIP_XCD_MODE")
    probs, _ = _device_circuits(other, N, K, B, 4000)
    run_batch(probs, pri, **opts)
    for c, q in enumerate(probs):
        _assert_same_fit(q.fetch(), ref[c], "xcd mode %d circuit %d" % (mode, c))
        q.close()
    other.close()


def test_fit_batch_128_as_4x32_streams(ctx):
    """fit_batch(streams=4): 128 circuits of the headline shape as 4 lock-step batches of 32 on four contexts (HIP streams + mailboxes),
    one host thread each -- and each of those batches as the library's two interleaved halves on its context's twin stream (eight
    batches in flight, four helper threads): every circuit bit-identical to its own single fit."""
    from circuitmap_amd import batch
    from circuitmap_amd.optimise.caviar import Problem
    from circuitmap_amd.simulation import simulate_device
    N, K, B = 1000, 2000, 128
    pri = default_priors(N)
    opts = dict(iters=50, seed=0)
    circuits = []
    for c in range(B):
        d = simulate_device(N=N, T=900, H=10, trials=K, seed=2000 + c, full=False, ctx=ctx)
        circuits.append((d["psc"], d["stim_matrix"]))
    got = batch.fit_batch(circuits, priors=pri, fit_options=opts, circuits_per_launch=32, streams=4)
    assert len(got) == B
    for c, (psc, stim) in enumerate(circuits):
        q = Problem(ctx, psc, stim)
        q.run(*prior_args(pri), **opts)
        res = q.fetch(want_lam=False)
        q.close()
        for i, nm in enumerate(OUT_NAMES):
            if nm != "lam":
                np.testing.assert_array_equal(np.asarray(res[i], dtype=float), np.asarray(got[c][nm], dtype=float), err_msg="circuit %d %s" % (c, nm))


@pytest.mark.parametrize("N,K,B,iters", [(200, 400, 8, 6), (330, 700, 3, 5), (1000, 2000, 8, 6), (1700, 3400, 1, 3)])
def test_cholesky_families_bitwise_equal(ctx, monkeypatch, N, K, B, iters):
    """The grouped Cholesky family (four panels per pass over the trailing matrix, 64 x 64 super-tiles, next diagonal block factored by
    workgroup 0 of the pass; default for batches and above 48 tile rows) leaves bitwise the factor of the panel-pair family
    (CAVIAR_HIP_CHOL_FAMILY=0): every tile product is its own matrix-core accumulation and the subtractions happen in panel order.
    Shapes: 7 tile rows (one full group + 3), 11 (odd super-row), 32 (the bench shape), 54 (one circuit above CM_LARGE_TILES)."""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import run_batch
    pri = default_priors(N)
    opts = dict(iters=iters, seed=0, fn_scan=False)
    res = []
    for fam in (2, 0):                                    # 2 = grouped also for one circuit (the default keeps the pair family there)
        monkeypatch.setenv("CAVIAR_HIP_CHOL_FAMILY", str(fam))
        cx = _lib.Context(0)
        monkeypatch.delenv("CAVIAR_HIP_CHOL_FAMILY")
        probs, _ = _device_circuits(cx, N, K, B, 5000)
        if B == 1:
            probs[0].run(*prior_args(pri), **opts)
        else:
            run_batch(probs, pri, **opts)
        res.append([q.fetch() for q in probs])
        for q in probs:
            q.close()
        cx.close()
    for c in range(B):
        _assert_same_fit(res[0][c], res[1][c], "grouped vs pair family, circuit %d" % c)
    assert np.count_nonzero(res[0][0][0]) > 0


def test_device_fn_scan_vs_oracle_random(ctx):
    """reconnect_spont_cells on the device (cm_fnscan.hip, one workgroup per circuit; cm_k_reconnect runs the kernel a fit runs): random
    scenarios against the oracle's literal restatement of caviar.py:102-144 -- several reconnections per case, ties in the event counts
    (first maximum in cell order), single-event reconnections (sem = NaN, quirk 7), minimum_spike_count 1..3, one to three powers,
    long rows -- mu, lam, z bit for bit, beta to 1e-14."""
    import warnings
    from circuitmap_amd import _lib
    from oracle import caviar_oracle as co
    rng = np.random.default_rng(11)
    for trial in range(14):
        N, K = int(rng.integers(3, 40)), int(rng.integers(12, 300))
        H = int(rng.integers(1, min(N, 6) + 1))
        powers = np.array([30.0, 50.0, 70.0])[: int(rng.integers(1, 4))]
        stim = np.zeros((N, K))
        for k in range(K):
            stim[rng.choice(N, H, replace=False), k] = powers[rng.integers(0, len(powers))]
        lam = rng.random((N, K)) * (stim > 0)
        mu = rng.normal(5, 2, N) * (rng.random(N) > 0.6)
        z = rng.uniform(2, 6, K) * (rng.random(K) > (0.4 if trial % 2 else 0.85))
        mc = float(rng.integers(1, 4))
        minimax = 0.2
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = co.reconnect_spont_cells(np.zeros(K), stim, lam, mu, rng.random(N) * 0 + 0.5, z, minimax, mc)
        gl, gm, gb, gz = _lib.f64(lam).copy(), _lib.f64(mu).copy(), 0.5 * np.ones(N), _lib.f64(z).copy()
        _lib.check(ctx.lib.cm_k_reconnect(ctx.handle, N, K, _lib.dptr(_lib.f64(stim)), _lib.dptr(gl), _lib.dptr(gm), _lib.dptr(gb), _lib.dptr(gz), minimax, mc))
        assert np.array_equal(gm, want[0]) and np.array_equal(gl, want[2]) and np.array_equal(gz, want[3]), trial
        assert np.allclose(gb, want[1], rtol=1e-14, atol=0, equal_nan=True), trial


def test_device_fn_scan_on_global_arrays():
    """Shapes whose scan state does not fit LDS (N = 5000 / K = 10000) run k_fn_scan on the global arrays when they come as a batch of eight
    circuits or more.  CAVIAR_HIP_FN_SCAN_GLOBAL=1 selects that instantiation for every shape: the random reconnect_spont_cells scenarios of
    test_device_fn_scan_vs_oracle_random (oracle: caviar.py:102-144, bit for bit) and a fit with reconnections, in ONE child process (the flag is
    read once per process)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, CAVIAR_HIP_FN_SCAN_GLOBAL="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_device_fn_scan_vs_oracle_random or test_caviar_vs_golden"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and " passed" in out, out[-3000:]


def test_fit_lockstep_accepts_scattered_device_circuits():
    """What a rank holds after fit_batch(scatter=True) over RCCL: float64 CUDA tensors [psc | stim] that never existed on its host.  They go into
    cm_problem_create(CM_MEM_DEVICE) by address; the fit equals the one from host arrays bit for bit."""
    import torch
    from circuitmap_amd import batch
    from circuitmap_amd.optimise import caviar
    N, K, T = 40, 120, 900
    pri = default_priors(N)
    opts = dict(iters=6, seed=1)
    sims = [_sim_case(N, K, 60 + c)[0] for c in range(3)]
    circuits = {}
    for c, sim in enumerate(sims):
        flat = np.concatenate([sim["psc"].ravel(), sim["stim_matrix"].ravel()])
        circuits[c] = ("device", torch.from_numpy(flat).cuda(), N, K, T)
    torch.cuda.synchronize()
    got = batch._fit_lockstep(circuits, [0, 1, 2], pri, opts, 0, 8, streams=1)
    for sim, g in zip(sims, got):
        res = caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), **opts)
        for i, nm in enumerate(OUT_NAMES):
            if nm != "lam":
                np.testing.assert_array_equal(np.asarray(res[i], dtype=float), np.asarray(g[nm], dtype=float), err_msg=nm)


# ---------------------------------------------------------------------------------------------------------------------------------
# Two builds / two environments of the library, compared through child processes (tests/child_fit.py)
# ---------------------------------------------------------------------------------------------------------------------------------
def _worst_rel(a, b, per_iteration):
    """largest relative deviation; histories are scaled iteration by iteration"""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if not per_iteration:
        return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)) if a.size else 0.0
    return max(float(np.max(np.abs(x - y)) / max(np.max(np.abs(y)), 1e-300)) for x, y in zip(a, b))


def test_mc_logit_residue_threshold(tmp_path):
    """The default build takes log(f / (1 - f)), f = sigmoid(x), as x itself for -700 < x < 8 (csrc/cm_bodies.h: CM_LOGIT_RESIDUE_FROM; the
    rounding residue it drops is <= 3.3e-13 there, reference caviar.py:233-236).  The variant build -DCM_LOGIT_RESIDUE_FROM=-1e300
    (circuitmap_amd/libcaviar_hip_litlogit.so, built by __graft_entry__.build(); it also switches k_mc_dead's screening off, whose bound
    assumes the threshold) evaluates the residue for every argument.  Four circuits of BASELINE configs[1]'s shape, all 50 iterations with
    histories, in a child process per build: trajectories agree to 1e-8 (mu, beta, z, rate per iteration, final lam; phi per iteration 1e-7) and NOT ONE gate
    decision differs (a neuron whose lam row is zero in one build and not in the other, at any iteration)."""
    import os
    from child_fit import fit_specs, run_child
    variant = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "circuitmap_amd", "libcaviar_hip_litlogit.so")
    assert os.path.exists(variant), "variant build missing: run __graft_entry__.build()"
    specs = [dict(N=300, K=1000, B=4, seed0=2000, batch=True, opts=dict(iters=50, seed=0, save_histories=True))]
    ours = fit_specs(specs)
    theirs = run_child(specs, {"CAVIAR_HIP_LIB": variant}, tmp_path)
    assert sorted(ours) == sorted(theirs)
    worst = {}
    for c in range(4):
        for nm in ("mu", "beta", "phi", "z", "rate"):
            key = "0_%d_hist_%s" % (c, nm)
            worst[nm] = max(worst.get(nm, 0.0), _worst_rel(ours[key], theirs[key], True))
        worst["lam"] = max(worst.get("lam", 0.0), _worst_rel(ours["0_%d_lam" % c], theirs["0_%d_lam" % c], False))
        assert np.array_equal(ours["0_%d_hist_alive" % c], theirs["0_%d_hist_alive" % c]), "gate flip in circuit %d" % c
        assert np.array_equal(ours["0_%d_mu" % c] != 0, theirs["0_%d_mu" % c] != 0)
    # measured (round 4, MI355X): mu 4.9e-10, beta 1.1e-9, lam 6.3e-11, phi 2.4e-8 (a row's Newton iteration amplifies the change of its lam row)
    assert all(worst[nm] <= 1e-8 for nm in ("mu", "beta", "z", "rate", "lam")) and worst["phi"] <= 1e-7, worst
    assert any(v > 0 for v in worst.values()), "the two builds cannot be bitwise equal: was the variant really loaded?"


def test_mc_dead_screening_adversarial(tmp_path):
    """k_mc_dead (csrc/cm_iter.hip) replaces the Monte-Carlo logit of an all-zero row by -inf when an UPPER BOUND of its mean logits already
    fails the gate of the next sweep (reference caviar.py:209-221: the logit only feeds that gate).  Adversarial inputs for the bound: per-neuron
    beta priors spread over 0.2 .. 10 (sigma beta^4 / 2 from negligible to decisive, so all-zero rows land on BOTH sides of the gate and some are
    revived), msrmp swept over 0.05 .. 0.7 so the bounds straddle the threshold, num_mc_samples 3 and 7 (the sample mean of the quantiles far
    from 0), delay_spont_est 0 and 5 (screening starts with the first gated sweep), and a phi prior (mean, var) = (0.5, 0.5) whose truncation
    point ndtr(-mean / sdev) is far from negligible (the bound must then refuse to decide).  Every circuit of every configuration, all outputs
    and all histories, bit-identical to the same batch with CAVIAR_HIP_NO_MC_SCREEN=1 (child process: the flag is read once)."""
    from child_fit import fit_specs, run_child
    specs = []
    for S in (3, 7):
        for delay in (0, 5):
            for phi, cov in ((None, None), ([0.5, 0.5], [0.5, 0.5])):
                for msrmp in (0.05, 0.2, 0.4, 0.7):
                    specs.append(dict(N=120, K=400, B=8, seed0=6000 + len(specs), batch=True, beta_seed=17 + len(specs), phi=phi, phi_cov_diag=cov,
                                      opts=dict(iters=9, seed=len(specs), num_mc_samples=S, delay_spont_est=delay, msrmp=msrmp, minimum_spike_count=2,
                                                save_histories=True)))
    ours = fit_specs(specs)
    theirs = run_child(specs, {"CAVIAR_HIP_NO_MC_SCREEN": "1"}, tmp_path)
    assert sorted(ours) == sorted(theirs)
    for k in ours:
        np.testing.assert_array_equal(ours[k], theirs[k], err_msg=k)
    # the inputs did exercise the gate on both sides in every regime: rows that the gated sweeps close (and that then stay all zero: the rows
    # k_mc_dead screens), rows that stay open next to them, configurations that end with nearly every row closed and with nearly every row open
    stayed, closed_frac = 0, []
    for si, spec in enumerate(specs):
        for c in range(spec["B"]):
            alive = ours["%d_%d_hist_alive" % (si, c)]
            first = spec["opts"]["delay_spont_est"] + 1                      # first gated sweep
            for it in range(max(first, 1), alive.shape[0]):
                stayed += int(np.sum(~alive[it - 1] & ~alive[it]))
            closed_frac.append(1.0 - alive[-1].mean())
    assert stayed > 1000 and min(closed_frac) < 0.3 and max(closed_frac) > 0.9, (stayed, min(closed_frac), max(closed_frac))
    # ... and the screened path is the reference's: circuit 0 of four configurations against the running-residual oracle on identical inputs
    # (one per regime: default / truncating phi prior, early / late gate)
    from child_fit import priors_of
    from circuitmap_amd.simulation import simulate_device
    from oracle import caviar_oracle as co
    for si in (1, 6, 10, 30):
        spec = specs[si]
        d = simulate_device(N=spec["N"], T=900, H=10, trials=spec["K"], seed=spec["seed0"], full=False)
        ref = co.caviar(d["psc"], d["stim_matrix"], *prior_args(priors_of(spec)), literal=False, **spec["opts"])
        for i, nm in enumerate(OUT_NAMES):
            assert rel_err(ours["%d_0_%s" % (si, nm)], ref[i]) < 1e-6, (si, nm)
        assert np.array_equal(ours["%d_0_hist_alive" % si], np.any(np.asarray(ref[11]) != 0, axis=2)), si


def test_rerun_more_iterations_fewer_samples(ctx):
    """The random tables of a problem are sized by (iters, N, S): a second run of the SAME problem objects with more iterations and fewer
    Monte-Carlo samples -- (50, 100) then (100, 50): the product iters x S does not grow -- must re-size the per-row table statistics
    (cm_pdev::wstat, iters x N rows; ADVICE r3: it was tied to the capacity of the sample table and overran into neighbouring pool slabs).
    Every circuit equals a fresh problem's (100, 50) run bit for bit."""
    from circuitmap_amd.optimise.caviar import run_batch
    pri = default_priors(120)
    probs, _ = _device_circuits(ctx, 120, 400, 4, 7100)
    run_batch(probs, pri, iters=50, num_mc_samples=100, seed=1)
    run_batch(probs, pri, iters=100, num_mc_samples=50, seed=1)
    again = [q.fetch() for q in probs]
    fresh, _ = _device_circuits(ctx, 120, 400, 4, 7100)
    run_batch(fresh, pri, iters=100, num_mc_samples=50, seed=1)
    for c, (q, a) in enumerate(zip(fresh, again)):
        _assert_same_fit(q.fetch(), a, "re-run circuit %d" % c)
    for q in probs + fresh:
        q.close()


def test_long_rows_par_chain_equals_single_consumer(tmp_path):
    """Rows of more than 64 targeted trials travel through the multi-wave ticketed chain (k_sweep_par) as ceil(len / 64) record segments.  The
    reference's published workload -- N = 200, K = 3600, H = 10: ~180 trials per cell (examples/demo_compressive_connectivity_mapping.ipynb:
    172-175) -- as single fits and as a lock-step batch, and a 9-power design whose long rows take the literal pool walk: every output and every
    per-iteration history bit-identical to the same fits walked by the single-consumer kernel (CAVIAR_HIP_NO_PAR_SWEEP=1, child process), which
    runs the same per-row function on the same records one row at a time: the tickets only change WHEN a row runs, never what it computes."""
    from child_fit import fit_specs, run_child
    opts = dict(iters=12, seed=0, save_histories=True)
    specs = [dict(N=200, K=3600, B=2, seed0=8100, batch=False, opts=opts, sim=dict(connection_prob=0.1)),
             dict(N=200, K=3600, B=8, seed0=8200, batch=True, opts=opts, sim=dict(connection_prob=0.1)),
             dict(N=60, K=1500, B=3, seed0=8300, batch=True, H=8, opts=dict(opts, msrmp=0.2), sim=dict(powers=[30.0 + 5 * i for i in range(9)], connection_prob=0.2))]
    ours = fit_specs(specs)                             # the library's own choice (dense chains: the single-consumer kernel, cm_par_class)
    forced = run_child(specs, {"CAVIAR_HIP_FORCE_PAR_SWEEP": "1"}, tmp_path)
    theirs = run_child(specs, {"CAVIAR_HIP_NO_PAR_SWEEP": "1"}, tmp_path)
    assert sorted(ours) == sorted(theirs) == sorted(forced)
    for k in ours:
        np.testing.assert_array_equal(forced[k], theirs[k], err_msg=k)
        np.testing.assert_array_equal(ours[k], theirs[k], err_msg=k)
    assert np.count_nonzero(ours["0_0_mu"]) > 0 and np.count_nonzero(ours["2_0_mu"]) > 0        # real fits: connections were found


def test_chain_tables_from_iter_mid_equal_presweep_scan(tmp_path):
    """Once every circuit of a set takes the in-LDS solve of k_iter_mid, block 0 of that kernel leaves the chain tables of the next sweep
    (ordered compaction of the rows with mu != 0, record offsets, chunk table: update_lam's sweep order, caviar.py:196-229) and the pre-sweep
    packs the records in a fourth role; iterations with a tiled solve keep the scan in the pre-sweep and k_seq_pack.  Same tables, same
    records: whole trajectories bitwise those of CAVIAR_HIP_NO_FUSED_PACK=1 (child process) -- one circuit and batches, a non-zero prior mean
    (every row a chain row: several scan rounds per workgroup, more rows than the packing role's grid), N not a multiple of the scan's four
    rows per thread, explicit histories, rows of more than 64 trials."""
    from child_fit import fit_specs, run_child
    opts = dict(iters=14, seed=0, save_histories=True)
    specs = [dict(N=300, K=1000, B=1, seed0=9600, batch=False, opts=opts),
             dict(N=301, K=700, B=6, seed0=9610, batch=True, opts=opts),
             dict(N=1000, K=2000, B=4, seed0=9620, batch=True, opts=dict(iters=12, seed=1)),
             dict(N=1203, K=900, B=2, seed0=9630, batch=True, beta_seed=3, mu_prior=0.5, opts=dict(iters=8, seed=2)),
             dict(N=200, K=3600, B=2, seed0=9640, batch=True, opts=dict(iters=10, seed=0), sim=dict(connection_prob=0.1))]
    ours = fit_specs(specs)
    theirs = run_child(specs, {"CAVIAR_HIP_NO_FUSED_PACK": "1"}, tmp_path)
    assert sorted(ours) == sorted(theirs)
    for k in ours:
        np.testing.assert_array_equal(ours[k], theirs[k], err_msg=k)
    assert np.count_nonzero(ours["0_0_mu"]) > 0 and np.count_nonzero(ours["2_0_mu"]) > 0
    # the tickets of the multi-wave chain: ranked column by column in the pre-sweep while the chain is long (cm_col_rank16x2), entry by entry in
    # k_seq_pack otherwise (CAVIAR_HIP_NO_COLUMN_RANKS: always) -- a wrong ticket would reorder the updates of a trial's residual or stall the chain
    plain = run_child(specs, {"CAVIAR_HIP_NO_COLUMN_RANKS": "1", "CAVIAR_HIP_NO_FUSED_PACK": "1"}, tmp_path)
    for k in ours:
        np.testing.assert_array_equal(ours[k], plain[k], err_msg=k)


def test_inverse_supertiles_bitwise_equal_one_tile(tmp_path):
    """The inverse of the Cholesky factor runs its recursive-doubling products (T = L21 X11, X21 = -X22 T; block_update_mu, caviar.py:170) on
    64 x 64 super-tiles from level b = 2 on.  Every 16 x 16 block accumulates the same products in the same order as in the one-tile kernels, so
    mu and beta -- and with them whole trajectories -- are bitwise those of CAVIAR_HIP_INV_ONE_TILE=1 (child process): odd and even tile
    counts (7, 11, 32 tile rows at the first solves), a single circuit and batches."""
    from child_fit import fit_specs, run_child
    specs = [dict(N=200, K=400, B=8, seed0=9100, batch=True, opts=dict(iters=6, seed=0)),
             dict(N=330, K=700, B=3, seed0=9200, batch=True, opts=dict(iters=5, seed=0)),
             dict(N=1000, K=2000, B=8, seed0=9300, batch=True, opts=dict(iters=5, seed=0)),
             dict(N=1000, K=2000, B=1, seed0=9400, batch=False, opts=dict(iters=4, seed=0)),
             dict(N=1700, K=3400, B=1, seed0=9500, batch=False, opts=dict(iters=3, seed=0))]
    ours = fit_specs(specs)
    theirs = run_child(specs, {"CAVIAR_HIP_INV_ONE_TILE": "1"}, tmp_path)
    assert sorted(ours) == sorted(theirs)
    for k in ours:
        np.testing.assert_array_equal(ours[k], theirs[k], err_msg=k)
