#!/usr/bin/env python3
"""Trajectory comparison of two builds of libcaviar_hip.so (same ABI), e.g. the default build against -DCM_LOGIT_RESIDUE_FROM=-1e300:
    python tools/compare_builds.py dump out_a.npz [cfg2|cfg3] [n_circuits]          (CAVIAR_HIP_LIB selects the build)
    python tools/compare_builds.py diff out_a.npz out_b.npz
`dump` fits seeded device-generated circuits with save_histories and stores mu / lam-row support / phi / z per iteration; `diff` reports the
largest relative deviation per output over all iterations and the number of gate flips (a neuron whose lam row is zero in one build and
not in the other, at any iteration)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {"cfg2": (300, 1000), "cfg3": (1000, 2000)}


def dump(path, cfg, n):
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem
    from circuitmap_amd.simulation import simulate_device
    N, K = SHAPES[cfg]
    ctx = _lib.default_context(0)
    pri = dict(mu=np.zeros(N), beta=1e1 * np.ones(N), shape=1.0, rate=1e-1, phi=np.c_[1e-1 * np.ones(N), 5e0 * np.ones(N)],
               phi_cov=np.array([np.array([[1e-1, 0], [0, 1e0]]) for _ in range(N)]))
    out = {}
    for c in range(n):
        d = simulate_device(N=N, T=900, H=10, trials=K, seed=2000 + c, keep_on_device=True, ctx=ctx)
        q = Problem(ctx, None, None, device_ptrs=(N, K, 900, d["psc_ptr"], d["stim_ptr"]))
        ctx.free(d["psc_ptr"]); ctx.free(d["stim_ptr"])
        q.run(pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"], pri["phi_cov"], iters=50, seed=0, save_histories=True)
        r = q.fetch()
        q.close()
        out["mu_%d" % c] = r[9]; out["alive_%d" % c] = np.any(r[11] != 0, axis=2); out["lam_%d" % c] = r[2]
        out["phi_%d" % c] = r[14]; out["z_%d" % c] = r[16]; out["final_mu_%d" % c] = r[0]; out["beta_%d" % c] = r[10]
    np.savez_compressed(path, n=n, **out)
    print("wrote", path)


def diff(a, b):
    A, B = np.load(a), np.load(b)
    n = int(A["n"])
    worst, flips, flip_its = {}, 0, []
    for c in range(n):
        for key in ("mu", "phi", "z", "beta", "final_mu", "lam"):
            x, y = A["%s_%d" % (key, c)], B["%s_%d" % (key, c)]
            if key in ("mu", "phi", "z", "beta"):          # per iteration: scale of that iteration
                for it in range(x.shape[0]):
                    sc = max(np.max(np.abs(y[it])), 1e-300)
                    worst[key] = max(worst.get(key, 0.0), float(np.max(np.abs(x[it] - y[it])) / sc))
            else:
                sc = max(np.max(np.abs(y)), 1e-300)
                worst[key] = max(worst.get(key, 0.0), float(np.max(np.abs(x - y)) / sc))
        fa, fb = A["alive_%d" % c], B["alive_%d" % c]
        d = fa != fb
        flips += int(d.sum())
        flip_its += [int(i) for i in np.nonzero(d.any(axis=1))[0]]
    print("circuits %d; largest relative deviation over all iterations: %s; gate flips (neuron x iteration): %d%s" % (
        n, ", ".join("%s %.2e" % kv for kv in sorted(worst.items())), flips, (" at iterations %s" % sorted(set(flip_its))) if flips else ""))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "cfg3", int(sys.argv[4]) if len(sys.argv) > 4 else 4)
    else:
        diff(sys.argv[2], sys.argv[3])
