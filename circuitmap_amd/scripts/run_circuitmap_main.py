#!/usr/bin/env python3
"""MI355X counterpart of the reference's pipeline driver (scripts/run_circuitmap_main.py:14-63), same command line, same
input and output files:

    python -m circuitmap_amd.scripts.run_circuitmap_main --data expt.mat --config cfg.yaml --out results/

* `--data`    `.mat` (scipy.io) or `.npy/.npz` holding `psc` [K, T] and `stimulus_matrix` [N, K]        (reference :22-32)
* `--config`  YAML with `demixer` (checkpoint path) and `msrmp`                                           (reference :35-42)
* `--out`     directory; writes `<stem>_cmap.mat` and `<stem>_cmap.npz` with `weights` (mu), `weight_uncertainty`
              (beta), `spikes` (lam)                                                                      (reference :51-63)

Both compute stages run on the GPU: the demixer (include/nwd_hip.h) and CAVIaR (include/caviar_hip.h).  The reference
pins the demixer to the CPU "to avoid memory clash between pytorch and JAX" (:38); here both stages share one HIP context.
"""
import argparse
import os
from pathlib import Path

import numpy as np


def load_data(path):
    """`psc`, `stimulus_matrix` from a .mat / .npy / .npz file (reference :22-32; any other extension raises, as there)."""
    ext = path[-4:]
    if ext == '.mat':
        from scipy.io import loadmat
        f = loadmat(path)
    elif ext == '.npy' or ext == '.npz':
        f = np.load(path, allow_pickle=False)
    else:
        raise Exception
    return np.asarray(f['psc'], dtype=np.float64), np.asarray(f['stimulus_matrix'], dtype=np.float64)


def save_results(out, data_path, state):
    """`<out>/<stem>_cmap.mat` and `.npz` (reference :51-63)"""
    from scipy.io import savemat
    if out[-1] != '/':
        out += '/'
    save_name = out + Path(data_path).stem + '_cmap'
    res = {'weights': state['mu'], 'weight_uncertainty': state['beta'], 'spikes': state['lam']}
    savemat(save_name + '.mat', res)
    np.savez(save_name, **res)
    return save_name


def run(data, config, out, demixer_state=None, fit_options=None, device=0, verbose=True):
    """The body of the reference script as a function.  `demixer_state` (dict of parameter arrays) overrides the checkpoint
    named in the config -- used by the tests, which have no trained checkpoint."""
    import yaml
    import circuitmap_amd as cm
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    psc, stim_matrix = load_data(data)
    cfg = yaml.safe_load(open(config))
    demix = NeuralDemixer(path=None if demixer_state is not None else cfg['demixer'], state=demixer_state, device=device)
    psc_dem = demix(psc, verbose=verbose)
    msrmp = float(cfg['msrmp'])
    N = stim_matrix.shape[0]
    model = cm.Model(N)
    opts = {'msrmp': msrmp, 'save_histories': False}
    opts.update(fit_options or {})
    model.fit(psc_dem, stim_matrix, method='caviar', fit_options=opts)
    os.makedirs(out, exist_ok=True)
    return model, save_results(out, data, model.state)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--data')
    parser.add_argument('--config')
    parser.add_argument('--out')
    args = parser.parse_args(argv)
    run(args.data, args.config, args.out)


if __name__ == '__main__':
    main()
