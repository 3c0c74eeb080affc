"""Worker of the GPU parity tests that compare two BUILDS or two ENVIRONMENTS of the library (both are read once per process):

    python tests/child_fit.py spec.json out.npz      (CAVIAR_HIP_LIB / CAVIAR_HIP_NO_MC_SCREEN / ... set by the parent)

`fit_specs(specs)` is also what the parent runs in its own process for its side of the comparison, so both sides execute the same
Python.  A spec: N, K, B, seed0, opts (fit options), batch (lock-step batch or one fit per circuit), and the prior variant:
beta_seed (per-neuron beta prior 10^U(-0.7, 1); None: the Model default), mu_prior (constant prior mean), phi / phi_cov_diag (prior mean and covariance diagonal); sim: further
keyword arguments of the device generator (powers, connection_prob, ...).
Test infrastructure only: nothing here is imported by the product."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def priors_of(spec):
    from helpers import default_priors
    N = spec["N"]
    pri = default_priors(N)
    if spec.get("beta_seed") is not None:
        pri["beta"] = 10.0 ** np.random.default_rng(spec["beta_seed"]).uniform(-0.7, 1.0, N)
    if spec.get("mu_prior") is not None:
        pri["mu"] = np.full(N, float(spec["mu_prior"]))
    if spec.get("phi") is not None:
        pri["phi"] = np.tile(np.asarray(spec["phi"], dtype=float), (N, 1))
    if spec.get("phi_cov_diag") is not None:
        d = np.asarray(spec["phi_cov_diag"], dtype=float)
        pri["phi_cov"] = np.array([np.diag(d) for _ in range(N)])
    return pri


def fit_specs(specs):
    """-> {"<spec index>_<circuit>_<name>": array}: the eight outputs (+ histories with save_histories; lam histories as the row-support
    mask [iters, N] and the per-row sums, which is what the comparisons need) of every circuit of every spec"""
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem, run_batch
    from circuitmap_amd.simulation import simulate_device
    from helpers import OUT_NAMES, prior_args
    ctx = _lib.default_context(0)
    out = {}
    for si, spec in enumerate(specs):
        N, K, B, T = spec["N"], spec["K"], spec["B"], spec.get("T", 900)
        pri = priors_of(spec)
        probs = []
        for c in range(B):
            d = simulate_device(N=N, T=T, H=spec.get("H", 10), trials=K, seed=spec["seed0"] + c, keep_on_device=True, ctx=ctx, **spec.get("sim", {}))
            probs.append(Problem(ctx, None, None, device_ptrs=(N, K, T, d["psc_ptr"], d["stim_ptr"])))
            ctx.free(d["psc_ptr"]); ctx.free(d["stim_ptr"])
        opts = dict(spec["opts"])
        if spec.get("batch", True):
            run_batch(probs, pri, interleave=False, **opts)
        else:
            for q in probs:
                q.run(*prior_args(pri), **opts)
        for c, q in enumerate(probs):
            r = q.fetch()
            for i, nm in enumerate(OUT_NAMES):
                out["%d_%d_%s" % (si, c, nm)] = np.asarray(r[i], dtype=float)
                if opts.get("save_histories"):
                    h = np.asarray(r[9 + i], dtype=float)
                    if nm == "lam":
                        out["%d_%d_hist_alive" % (si, c)] = np.any(h != 0, axis=2)
                        out["%d_%d_hist_lamsum" % (si, c)] = h.sum(axis=2)
                    elif nm in ("shape", "rate"):
                        out["%d_%d_hist_%s" % (si, c, nm)] = h[:, 0].copy()
                    else:
                        out["%d_%d_hist_%s" % (si, c, nm)] = h
            q.close()
    return out


def run_child(specs, env_extra, tmp_path, timeout=900):
    """fit_specs(specs) in a child process with `env_extra` added to the environment -> the loaded result dict"""
    import subprocess
    spec_file, out_file = os.path.join(str(tmp_path), "spec.json"), os.path.join(str(tmp_path), "out.npz")
    with open(spec_file, "w") as f:
        json.dump(specs, f)
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), spec_file, out_file], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    with np.load(out_file) as z:
        return {k: z[k] for k in z.files}


if __name__ == "__main__":
    import torch  # noqa: F401  (first: one HIP runtime per process)
    with open(sys.argv[1]) as f:
        specs_in = json.load(f)
    np.savez(sys.argv[2], **fit_specs(specs_in))
