"""Generator oracle (oracle/sim_oracle.py): structural contract of the reference's simulate (circuitmap/simulation.py:25-195)
and distributional agreement with the NumPy restatement circuitmap_amd.simulation.simulate (CPU, no GPU)."""
import numpy as np
import pytest

from oracle import sim_oracle as so


@pytest.fixture(scope="module")
def sim():
    return so.simulate(N=120, trials=720, seed=11)


def test_design_contract(sim):
    st = sim["stim_matrix"]
    N, K = st.shape
    assert (N, K) == (120, 720)
    assert np.all((st > 0).sum(axis=0) == 10)                              # H spots per hologram (:52)
    assert set(np.unique(st)) == {0.0, 45.0, 55.0, 65.0}
    assert np.array_equal(sim["I"], st.max(axis=0))                        # one power per trial (:79)
    # blockwise: in every block each neuron is hit once per power -> 720 / (3 * 12) = 20 blocks... equal counts per neuron and power
    for p in (45.0, 55.0, 65.0):
        assert np.all((st == p).sum(axis=1) == 720 // (3 * 12))
    tc = sim["trial_cells"]
    for k in (0, 5, 719):
        assert sorted(tc[k]) == sorted(np.flatnonzero(st[:, k]))


def test_population_contract(sim):
    w = sim["weights"]
    n_conn = int(0.05 * 120)
    assert np.count_nonzero(w) == n_conn
    strong = np.sort(w[w > 0])[::-1][:int(np.ceil(0.2 * n_conn))]
    assert np.all((strong >= 20) & (strong <= 40)) and np.all(w[w > 0] >= 9)
    assert np.all((sim["phi_0"] >= 0.2) & (sim["phi_0"] <= 0.25)) and np.all((sim["phi_1"] >= 10) & (sim["phi_1"] <= 15))
    assert np.all((sim["tau_r"] >= 25) & (sim["tau_r"] <= 60)) and np.all(sim["tau_d"] - sim["tau_r"] >= 75)
    sp, st = sim["spks"], sim["stim_matrix"]
    assert np.all(sp[st == 0] == 0) and set(np.unique(sp)) <= {0.0, 1.0}
    for n in range(120):                                                   # minimum rate at maximal power (:98-108)
        assert sp[n, st[n] == 65].mean() >= 0.4 - 1e-12
    t = sim["spk_times"]
    assert np.all(t[sp == 1] > 160) and np.all(t[sp == 0] == 0)
    assert np.all(np.isfinite(sim["psc"]))


def test_counter_based_draws_are_reproducible_and_independent_of_size():
    a = so.simulate(N=60, trials=240, seed=3)
    b = so.simulate(N=60, trials=240, seed=3)
    for k in ("psc", "stim_matrix", "weights", "spks"):
        np.testing.assert_array_equal(a[k], b[k])
    c = so.simulate(N=60, trials=240, seed=4)
    assert not np.array_equal(a["stim_matrix"], c["stim_matrix"])
    u = so.uniform(5, so.S_SPIKE, np.arange(200000, dtype=np.uint64))
    assert abs(u.mean() - 0.5) < 3e-3 and abs(u.var() - 1 / 12) < 1e-3 and u.min() >= 0 and u.max() < 1
    z = so.normal(5, so.S_IID, np.arange(200000, dtype=np.uint64))
    assert abs(z.mean()) < 1e-2 and abs(z.var() - 1) < 1e-2
    g = so.gamma(5, np.arange(100000, dtype=np.uint64), np.full(100000, 1e4 / 55.0 ** 2), 15.0)
    a_ = 1e4 / 55.0 ** 2
    assert abs(g.mean() / (a_ * 15.0) - 1) < 1e-2 and abs(g.var() / (a_ * 225.0) - 1) < 3e-2
    gb = so.gamma(6, np.arange(100000, dtype=np.uint64), np.full(100000, 0.6), 2.0)          # shape < 1: boosted branch
    assert abs(gb.mean() / 1.2 - 1) < 2e-2
    p = so.random_order(7, so.S_ORDER, np.uint64(0), 1000)
    assert sorted(p) == list(range(1000))


def test_same_distribution_as_numpy_restatement():
    """moments of the generated experiment against circuitmap_amd.simulation.simulate (the MT-free NumPy restatement of the
    same generative model): different draws, same distributions"""
    from circuitmap_amd.simulation import simulate
    a = so.simulate(N=300, trials=1500, seed=1)
    b = simulate(N=300, trials=1500, seed=1)
    assert abs(a["spks"].mean() / b["spks"].mean() - 1) < 0.2           # 300 sampled power curves each: +-10 % between seeds
    ta, tb = a["spk_times"][a["spks"] == 1], b["spk_times"][b["spks"] == 1]
    assert abs(ta.mean() / tb.mean() - 1) < 0.02 and abs(ta.std() / tb.std() - 1) < 0.1
    ga, gb = a["gp_noise"], b["gp_noise"]
    assert abs(ga.std() / gb.std() - 1) < 0.05
    ca = np.mean(ga[:, :-50] * ga[:, 50:]) / ga.var()
    assert abs(ca - np.exp(-0.5)) < 0.03                                   # squared-exponential kernel at lag = lengthscale
    na = a["psc"] - a["gp_noise"] - a["spont_pscs"]
    assert abs((a["spont_pscs"].max(axis=1) > 0).mean() - 0.05) < 0.02     # spontaneous-event probability
    # evoked amplitude per spike of a connected cell: weight x unit-area kernel -> trace integral ~ sum of weights that fired
    ya = np.sum(na, axis=1)
    wa = (a["spks"] * a["weights"][:, None]).sum(axis=0)
    assert np.corrcoef(ya, wa)[0, 1] > 0.98


def test_random_design_contract():
    """design='random' (:65-77): equal power representation, H targets per trial drawn with replacement"""
    a = so.simulate(N=40, trials=301, design="random", seed=4)
    st = a["stim_matrix"]
    nt = (st > 0).sum(axis=0)
    assert nt.max() <= 10 and nt.min() >= 1 and (nt < 10).any()            # repeated draws collapse
    counts = sorted((a["I"] == p).sum() for p in (45.0, 55.0, 65.0))
    assert counts == [100, 100, 101]                                       # np.array_split sizes
    assert np.array_equal(a["I"], st.max(axis=0))
    for k in range(301):
        c = a["trial_cells"][k]
        assert sorted(c[c >= 0]) == sorted(np.flatnonzero(st[:, k]))
