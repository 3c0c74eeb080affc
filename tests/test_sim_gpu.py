"""Device-side experiment generator (include/sim_hip.h) against its oracle (oracle/sim_oracle.py): integer structure bit for
bit, floating-point fields to rounding; full-size properties; and the generated experiment feeds CAVIaR from HBM."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    s = max(np.max(np.abs(b)), 1e-300)
    return np.max(np.abs(a - b)) / s < tol


@pytest.mark.parametrize("kw", [dict(N=60, trials=240, seed=3), dict(N=97, trials=500, H=7, seed=8, powers=(30.0, 50.0, 70.0, 120.0)),
                                dict(N=300, trials=1000, seed=1), dict(N=64, trials=200, nreps=2, T=700, seed=5, gp_lengthscale=25.0),
                                dict(N=80, trials=301, design="random", seed=9), dict(N=12, trials=90, H=10, design="random", seed=2)])
def test_matches_oracle(kw):
    from oracle import sim_oracle as so
    from circuitmap_amd.simulation import simulate_device
    okw = dict(kw)
    dev = simulate_device(**kw)
    ref = so.simulate(**okw)
    np.testing.assert_array_equal(dev["stim_matrix"], ref["stim_matrix"])          # design: bit for bit
    np.testing.assert_array_equal(dev["I"], ref["I"])
    np.testing.assert_array_equal(dev["spks"], ref["spks"])                        # Bernoulli draws + padding: bit for bit
    np.testing.assert_array_equal(dev["weights"] != 0, ref["weights"] != 0)
    for k in ("weights", "phi_0", "phi_1", "tau_r", "tau_d"):
        assert _close(dev[k], ref[k], 1e-13), k
    assert np.array_equal(np.floor(dev["spk_times"]), np.floor(ref["spk_times"]))  # the integer latency that places the PSC
    assert _close(dev["spk_times"], ref["spk_times"], 1e-12)
    assert _close(dev["gp_noise"], ref["gp_noise"], 1e-11)
    assert _close(dev["spont_pscs"], ref["spont_pscs"], 1e-11)
    assert _close(dev["psc"], ref["psc"], 1e-11)


def test_full_size_properties_and_feeds_caviar():
    """cfg3 shape generated in HBM and fitted without a host round trip of psc / stim."""
    import ctypes as C
    from circuitmap_amd import _lib
    from circuitmap_amd.simulation import simulate_device
    from circuitmap_amd.optimise.caviar import Problem
    from tests.helpers import default_priors, prior_args
    N, K, T = 1000, 2000, 900
    d = simulate_device(N=N, trials=K, seed=2000, keep_on_device=True)
    ctx = d["ctx"]
    assert d["device_ms"] > 0
    stim = ctx.download(d["stim_ptr"], np.empty((N, K)))
    psc = ctx.download(d["psc_ptr"], np.empty((K, T)))
    assert np.all((stim > 0).sum(axis=0) == 10) and set(np.unique(stim)) == {0.0, 45.0, 55.0, 65.0}
    assert np.array_equal(d["I"], stim.max(axis=0))
    assert np.count_nonzero(d["weights"]) == 50 and np.all(np.isfinite(psc))
    prob = Problem(ctx, None, None, device_ptrs=(N, K, T, d["psc_ptr"], d["stim_ptr"]))
    pri = default_priors(N)
    prob.run(*prior_args(pri), iters=50, seed=0)
    mu = prob.fetch()[0]
    assert np.corrcoef(d["weights"], mu)[0, 1] ** 2 > 0.6
    # same problem from host copies of the generated arrays: identical result
    prob2 = Problem(ctx, psc, stim)
    prob2.run(*prior_args(pri), iters=50, seed=0)
    np.testing.assert_array_equal(prob2.fetch()[0], mu)
    prob.close(); prob2.close()
    ctx.free(d["psc_ptr"]); ctx.free(d["stim_ptr"])


def test_argument_checks():
    from circuitmap_amd import _lib
    from circuitmap_amd.simulation import simulate_device
    with pytest.raises(TypeError):
        simulate_device(N=50, trials=100, bogus=1)
    with pytest.raises(AssertionError):
        simulate_device(N=50, trials=100, design="spiral")
    with pytest.raises(_lib.CaviarHipError):
        simulate_device(N=50, trials=100, H=100)
    with pytest.raises(_lib.CaviarHipError):
        simulate_device(N=20000, trials=100)
