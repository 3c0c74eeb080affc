"""The reference's pipeline driver (scripts/run_circuitmap_main.py) on the MI355X path: file formats on the CPU, the whole
pipeline (files -> demixer -> CAVIaR -> files) against the two oracles on the GPU."""
import os

import numpy as np
import pytest


def _make_inputs(tmp_path, fmt, N=40, K=160, seed=5):
    import circuitmap_amd as cm
    sim = cm.simulate(N=N, trials=K, seed=seed)
    data = str(tmp_path / ("expt." + fmt))
    if fmt == "mat":
        from scipy.io import savemat
        savemat(data, {"psc": sim["psc"], "stimulus_matrix": sim["stim_matrix"]})
    else:
        np.savez(data, psc=sim["psc"], stimulus_matrix=sim["stim_matrix"])
    cfg = str(tmp_path / "cfg.yaml")
    with open(cfg, "w") as f:
        f.write("demixer: './demixers/none.ckpt'\nmsrmp: 0.4\n")
    return sim, data, cfg


@pytest.mark.parametrize("fmt", ["mat", "npz"])
def test_load_and_save_formats(tmp_path, fmt):
    from circuitmap_amd.scripts.run_circuitmap_main import load_data, save_results
    sim, data, _ = _make_inputs(tmp_path, fmt)
    psc, stim = load_data(data)
    np.testing.assert_array_equal(psc, sim["psc"])
    np.testing.assert_array_equal(stim, sim["stim_matrix"])
    state = {"mu": np.arange(4.0), "beta": np.ones(4), "lam": np.eye(4)}
    os.makedirs(tmp_path / "out", exist_ok=True)
    name = save_results(str(tmp_path / "out"), data, state)
    assert name.endswith("expt_cmap")
    from scipy.io import loadmat
    m, z = loadmat(name + ".mat"), np.load(name + ".npz")
    for key, ref in (("weights", state["mu"]), ("weight_uncertainty", state["beta"]), ("spikes", state["lam"])):
        np.testing.assert_array_equal(np.squeeze(m[key]), ref)
        np.testing.assert_array_equal(z[key], ref)
    with pytest.raises(Exception):
        load_data(str(tmp_path / "expt.h5x"))


@pytest.mark.gpu
def test_pipeline_matches_oracles(tmp_path):
    from oracle import caviar_oracle as co
    from oracle import nwd_oracle as no
    from circuitmap_amd.scripts.run_circuitmap_main import run
    from tests.helpers import default_priors
    sim, data, cfg = _make_inputs(tmp_path, "mat")
    state = no.random_state(seed=6)
    model, name = run(data, cfg, str(tmp_path / "res"), demixer_state=state, fit_options={"iters": 8}, verbose=False)
    # stage 1: the demixed traces agree with the float32 oracle to rounding
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    psc_dem = NeuralDemixer(state=state)(sim["psc"], verbose=False)                 # deterministic: same bits as inside run()
    o_dem = no.demix(sim["psc"], state)
    assert np.any(o_dem != 0) and np.max(np.abs(psc_dem - o_dem)) / np.max(np.abs(o_dem)) < 2e-5
    # stage 2: CAVIaR on those traces agrees with the float64 oracle on the same traces
    pri = default_priors(sim["stim_matrix"].shape[0])
    ref = co.caviar(psc_dem, sim["stim_matrix"], pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"], pri["phi_cov"],
                    literal=False, iters=8, msrmp=0.4, save_histories=False)
    z = np.load(name + ".npz")
    for key, r in (("weights", ref[0]), ("weight_uncertainty", ref[1]), ("spikes", ref[2])):
        err = np.max(np.abs(z[key] - np.asarray(r))) / max(np.max(np.abs(r)), 1e-300)
        assert err < 1e-6, (key, err)
    np.testing.assert_array_equal(z["weights"], model.state["mu"])
    from scipy.io import loadmat
    np.testing.assert_array_equal(np.squeeze(loadmat(name + ".mat")["spikes"]), z["spikes"])
