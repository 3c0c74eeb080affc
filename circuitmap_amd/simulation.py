"""Seeded synthetic mapping experiments (host side, NumPy).

Same generative model, argument names, defaults and returned keys as the reference's
``circuitmap.simulation.simulate`` (reference ``circuitmap/simulation.py:25-195``), written from
scratch around a ``numpy.random.Generator`` so that benchmark and parity inputs are reproducible
(the reference draws from the unseeded global ``np.random``).  The PSC synthesis is done per
spike (shift-and-add of the presynaptic cell's unit-area kernel) instead of the reference's
dense N x K x T vmapped convolution; the result is the same trace.
"""
import itertools

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz


def _psc_kernels(tau_r, tau_d, T, eps=1e-5):
    """Unit-area difference-of-exponentials kernels (reference simulation.py:285-289)."""
    kr = np.arange(T)[None, :]
    ke = np.exp(-kr / tau_d[:, None]) - np.exp(-kr / tau_r[:, None])
    return ke / (_trapz(ke, axis=-1)[:, None] + eps)


def _sample_gp(rng, trial_dur, n_samples, gp_scale, gp_lengthscale):
    """Squared-exponential GP noise (reference simulation.py:211-215)."""
    d = np.arange(trial_dur)
    Kmat = np.exp(-(d[:, None] - d[None, :]) ** 2 / (2 * gp_lengthscale ** 2))
    w, V = np.linalg.eigh(Kmat)
    root = V * np.sqrt(np.clip(w, 0.0, None))[None, :]
    return gp_scale * (rng.standard_normal((n_samples, trial_dur)) @ root.T)


def blockwise_design(rng, N, H, trials, nreps, powers):
    """H-spot holograms tiling a fresh random neuron order per block, highest power first
    (reference simulation.py:44-63)."""
    cols = []
    powers = np.sort(powers)[::-1]
    K = 0
    nholo = int(np.ceil(N / H))
    while K < trials:
        order = rng.permutation(N)
        holos = [order[i * H:(i + 1) * H] for i in range(nholo)]
        for power, holo, _ in itertools.product(powers, holos, range(nreps)):
            if K >= trials:
                break
            cols.append((holo, float(power)))
            K += 1
    reorder = rng.permutation(K)
    stim = np.zeros((N, K))
    for k_new, k_old in enumerate(reorder):
        holo, power = cols[k_old]
        stim[holo, k_new] = power
    return stim


def random_design(rng, N, H, trials, powers):
    """reference simulation.py:65-77 (targets drawn with replacement, equal power representation)."""
    K = trials
    stim = np.zeros((N, K))
    per = np.array_split(np.ones(K), len(powers))
    power_order = rng.permutation(np.concatenate([p * a for p, a in zip(powers, per)]))
    for k in range(K):
        stim[rng.integers(0, N, H), k] = power_order[k]
    return stim


def simulate(N=300, T=900, H=10, trials=1000, nreps=1, connection_prob=0.05, powers=(45, 55, 65),
             min_latency=160, gamma_beta=1.5e1, sigma=6e-4, frac_strongly_connected=0.2,
             strong_weight_lower=20, strong_weight_upper=40, weak_exp_mean=4, min_weight=9,
             phi_0_lower=0.2, phi_0_upper=0.25, phi_1_lower=10, phi_1_upper=15, mult_noise_log_var=0.01,
             tau_r_min=25, tau_r_max=60, tau_delta_min=75, tau_delta_max=250, weights=None,
             phi_0=None, phi_1=None, gp_scale=4e-3, gp_lengthscale=50, spont_prob=0.05,
             design="blockwise", max_power_min_spike_rate=0.4, seed=0, lite=False):
    """Returns the reference's ``sim`` dict (keys: weights, phi_0, phi_1, mult_noise, sigma, stim_matrix,
    psc, gp_noise, kernel, spks, spk_times, spont_pscs, I).

    ``lite=True`` skips the 900x900 GP draw for very large K (iid noise only)."""
    assert design in ["random", "blockwise"]
    rng = np.random.default_rng(seed)
    powers = np.asarray(powers, dtype=float)
    if design == "blockwise":
        stim_matrix = blockwise_design(rng, N, H, trials, nreps, powers)
    else:
        stim_matrix = random_design(rng, N, H, trials, powers)
    K = stim_matrix.shape[1]
    I = stim_matrix.max(axis=0)

    tau_r = rng.uniform(tau_r_min, tau_r_max, N)
    tau_d = tau_r + rng.uniform(tau_delta_min, tau_delta_max, N)
    kernels = _psc_kernels(tau_r, tau_d, T)

    if phi_0 is None or phi_1 is None:
        phi_0 = rng.uniform(phi_0_lower, phi_0_upper, N)
        phi_1 = rng.uniform(phi_1_lower, phi_1_upper, N)
    targeted = stim_matrix > 0
    frates = targeted / (1.0 + np.exp(-(phi_0[:, None] * stim_matrix - phi_1[:, None])))
    spks = (rng.random((N, K)) <= frates).astype(float) * targeted
    mult_noise = rng.lognormal(0, mult_noise_log_var, (N, K))

    # pad spikes so the rate at maximal power is at least max_power_min_spike_rate (simulation.py:98-108)
    max_power = powers.max()
    for n in range(N):
        locs = np.where(stim_matrix[n] == max_power)[0]
        if len(locs) == 0:
            continue
        fr_diff = max_power_min_spike_rate - np.mean(spks[n, locs])
        if fr_diff > 0:
            zero_locs = np.where(spks[n, locs] == 0)[0]
            req = min(int(np.ceil(fr_diff * len(locs))), len(zero_locs))
            spks[n, locs[rng.choice(zero_locs, req, replace=False)]] = 1.0

    nn, kk = np.nonzero(spks)
    spk_times = np.zeros((N, K))
    alpha = 1e4 / stim_matrix[nn, kk] ** 2
    spk_times[nn, kk] = min_latency + rng.gamma(alpha, gamma_beta)

    if weights is None:
        n_connected = int(connection_prob * N)
        connected = rng.choice(N, n_connected, replace=False)
        n_strong = int(np.ceil(frac_strongly_connected * n_connected))
        strong = rng.choice(connected, n_strong, replace=False)
        weak = np.setdiff1d(connected, strong)
        weights = np.zeros(N)
        weights[strong] = rng.uniform(strong_weight_lower, strong_weight_upper, n_strong)
        weights[weak] = rng.exponential(weak_exp_mean, len(weak)) + min_weight
    else:
        weights = np.asarray(weights, dtype=float)
        connected = np.flatnonzero(weights != 0)

    # evoked PSCs: shift-and-add of the presynaptic kernel at the spike time, renormalised to unit
    # area after truncation, times multiplicative noise and weight (simulation.py:17-23)
    psc = np.zeros((K, T))
    for n, k in zip(nn, kk):
        if weights[n] == 0:
            continue
        s = int(spk_times[n, k])
        if s >= T:
            continue
        ke = np.zeros(T)
        ke[s:] = kernels[n, :T - s]
        psc[k] += ke / (np.sum(ke) + 1e-5) * mult_noise[n, k] * weights[n]

    # spontaneous PSCs (simulation.py:157-170)
    spont_pscs = np.zeros((K, T))
    Trange = np.arange(T)
    wlo, whi = (np.min(weights[connected]), np.max(weights[connected])) if len(connected) else (0.0, 0.0)
    for k in np.flatnonzero(rng.random(K) <= spont_prob):
        tr = rng.uniform(tau_r_min, tau_r_max)
        td = tr + rng.uniform(tau_delta_min, tau_delta_max)
        t0 = rng.integers(1, T)
        kern = (np.exp(-(Trange - t0) / td) - np.exp(-(Trange - t0) / tr)) * (Trange > t0)
        spont_pscs[k] = rng.uniform(wlo, whi) * kern / (_trapz(kern) + 1e-5)

    gp_noise = np.zeros((K, T)) if lite else _sample_gp(rng, T, K, gp_scale, gp_lengthscale)
    noise = rng.normal(0, sigma, (K, T))
    psc = psc + spont_pscs + gp_noise + noise

    return {
        "weights": weights, "phi_0": phi_0, "phi_1": phi_1, "mult_noise": mult_noise, "sigma": sigma,
        "stim_matrix": stim_matrix, "psc": psc, "gp_noise": gp_noise, "kernel": None, "spks": spks,
        "spk_times": spk_times, "spont_pscs": spont_pscs, "I": I,
    }


_SIM_KEYS = ("N", "T", "H", "trials", "nreps", "connection_prob", "min_latency", "gamma_beta", "sigma", "frac_strongly_connected",
             "strong_weight_lower", "strong_weight_upper", "weak_exp_mean", "min_weight", "phi_0_lower", "phi_0_upper", "phi_1_lower",
             "phi_1_upper", "mult_noise_log_var", "tau_r_min", "tau_r_max", "tau_delta_min", "tau_delta_max", "gp_scale",
             "gp_lengthscale", "spont_prob", "max_power_min_spike_rate", "seed")


def simulate_device(powers=(45, 55, 65), design="blockwise", device_id=0, keep_on_device=False, full=True, ctx=None, **kw):
    """`simulate` generated ON the MI355X (include/sim_hip.h, circuitmap_amd/csrc/cm_sim.hip): same arguments, defaults and
    returned keys as the reference's ``simulate`` (circuitmap/simulation.py:25-195, both designs), counter-based draws.

    keep_on_device=False  -> dict of NumPy arrays (keys of the reference's `sim` dict; `full=False` skips the N x K and
                             per-component K x T extras and returns psc, stim_matrix, weights, phi_0, phi_1, I only)
    keep_on_device=True   -> `psc` and `stim_matrix` stay in HBM: dict with integer device addresses `psc_ptr`,
                             `stim_ptr` (free with ctx.free) for cm_problem_create(..., CM_MEM_DEVICE), plus the small host vectors.
    There is no CPU fallback; the NumPy restatement is `simulate` above."""
    import ctypes as C
    from . import _lib
    assert design in ["random", "blockwise"]
    dcode = 0 if design == "blockwise" else 1
    lib = _lib.load()
    o = _lib.CmSimOptions()
    lib.cm_sim_options_default(C.byref(o))
    unknown = set(kw) - set(_SIM_KEYS)
    if unknown:
        raise TypeError("simulate_device() got an unexpected keyword argument '%s'" % sorted(unknown)[0])
    for k, v in kw.items():
        setattr(o, k, type(getattr(o, k))(v))
    powers = [float(p) for p in powers]
    if not 1 <= len(powers) <= _lib.CM_SIM_MAX_POWERS:
        raise ValueError("1..%d powers" % _lib.CM_SIM_MAX_POWERS)
    o.n_powers = len(powers)
    for i, p in enumerate(powers):
        o.powers[i] = p
    ctx = ctx if ctx is not None else _lib.default_context(device_id)      # `ctx`: run on the caller's context / stream
    N, K, T = o.N, o.trials, o.T
    res = _lib.CmSimResult()
    ms = C.c_float(0.0)
    small = {k: np.empty(n) for k, n in (("weights", N), ("phi_0", N), ("phi_1", N), ("tau_r", N), ("tau_d", N), ("I", K))}
    if keep_on_device:
        psc_ptr, stim_ptr = ctx.alloc(K * T * 8), ctx.alloc(N * K * 8)
        dsmall = {k: ctx.alloc(v.nbytes) for k, v in small.items()}
        res.psc, res.stim_matrix = psc_ptr, stim_ptr
        for k, p in dsmall.items():
            setattr(res, k, p)
        try:
            with ctx.lock:
                _lib.check(lib.cm_simulate(ctx.handle, C.byref(o), dcode, C.byref(res), _lib.CM_MEM_DEVICE, C.byref(ms)))
            for k, v in small.items():
                ctx.download(dsmall[k], v)
        except Exception:
            ctx.free(psc_ptr); ctx.free(stim_ptr)
            raise
        finally:
            for p in dsmall.values():
                ctx.free(p)
        out = dict(small, psc_ptr=psc_ptr, stim_ptr=stim_ptr, N=N, K=K, T=T, sigma=o.sigma, device_ms=float(ms.value), ctx=ctx)
        return out
    big = {"psc": np.empty((K, T)), "stim_matrix": np.empty((N, K))}
    if full:
        big.update(spks=np.empty((N, K)), spk_times=np.empty((N, K)), gp_noise=np.empty((K, T)), spont_pscs=np.empty((K, T)))
    for k, v in list(small.items()) + list(big.items()):
        setattr(res, k, v.ctypes.data)
    with ctx.lock:
        _lib.check(lib.cm_simulate(ctx.handle, C.byref(o), dcode, C.byref(res), _lib.CM_MEM_HOST, C.byref(ms)))
    out = dict(small)
    out.update(big)
    out.update(sigma=o.sigma, kernel=None, mult_noise=None, device_ms=float(ms.value))
    return out
