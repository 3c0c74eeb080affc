"""Test infrastructure (a checker, like the rest of tests/): HIP path vs the running-residual oracle on BASELINE's configurations: max relative error of mu / lam (and every other output), gate
flips (rows whose lam support differs), for BASELINE.md section 6.  Needs an MI355X; the oracle runs on the host (about 2 minutes)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch  # noqa
from circuitmap_amd.optimise import caviar
from circuitmap_amd.simulation import simulate
from oracle import caviar_oracle as co
from tests.helpers import default_priors, prior_args, rel_err, OUT_NAMES
for name, N, K, iters, seed, lite in [("cfg2", 300, 1000, 50, 1300, False), ("cfg3", 1000, 2000, 50, 2000, False), ("cfg5", 5000, 10000, 3, 2005, True)]:
    sim = simulate(N=N, trials=K, seed=seed, lite=lite)
    pri = default_priors(N)
    res = caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), iters=iters, seed=0)
    t = time.time()
    ref = co.caviar(sim["psc"], sim["stim_matrix"], *prior_args(pri), literal=False, iters=iters, seed=0)
    dt = time.time() - t
    errs = {nm: rel_err(res[i], ref[i]) for i, nm in enumerate(OUT_NAMES)}
    flips = int(np.sum(np.any(res[2] != 0, axis=1) != np.any(ref[2] != 0, axis=1)))
    conn = int(np.sum((np.asarray(res[0]) != 0) != (np.asarray(ref[0]) != 0)))
    print("%s N=%d K=%d iters=%d: rel err mu %.2e lam %.2e (max over all outputs %.2e); rows with different lam support: %d; cells with different mu != 0: %d; "
          "oracle %.1f s = %.3f it/s" % (name, N, K, iters, errs["mu"], errs["lam"], max(errs.values()), flips, conn, dt, iters / dt), flush=True)
