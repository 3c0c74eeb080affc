"""CPU oracle for the device-side synthetic experiment generator (include/sim_hip.h, circuitmap_amd/csrc/cm_sim.hip).

TEST INFRASTRUCTURE ONLY: imported by tests/ and bench.py's checker legs, never by the product path.

What it restates.  The generative model of the reference's ``simulate`` (circuitmap/simulation.py:25-195, design =
'blockwise'): H-spot holograms tiling a fresh random neuron order per block, highest power first, shuffled (:44-63);
per-neuron PSC kernels and power curves (:81-92); Bernoulli spikes, padded to a minimum rate at maximal power
(:93-108); gamma latencies (:110-114, 197-198); 5 % connectivity, 20 % strong U(20,40) / weak 9+Exp(4) (:116-127);
evoked PSCs = shifted unit-area kernel, renormalised after truncation, x lognormal gain x weight (:17-23, 141-155);
spontaneous PSCs (:157-170); smooth correlated noise + iid noise (:172-176).

What it cannot restate bit for bit.  The reference draws from NumPy's global MT19937 stream in program order, which a
massively parallel generator cannot reproduce.  The device generator (and therefore this oracle) is COUNTER-BASED instead:
every variate is a pure function of (seed, stream, index) through Threefry-2x32-20 (oracle/threefry.py), so any thread
can produce any draw.  Distributions and structure follow the reference; the draws differ.  Two substitutions, both
distribution-preserving: random subsets/permutations are "sort by random key"; the squared-exponential GP noise
(exact multivariate normal in the reference) is white noise filtered with the Gaussian whose autocorrelation is that
kernel (exp(-d^2 / 2 l^2)), on a padded window, so it is stationary with the same covariance.
Parity: device == this oracle (integers exact, floats to rounding); contract vs the reference: tests/test_sim_*.py.
"""
import math

import numpy as np

from .threefry import threefry2x32

# stream ids
S_ORDER, S_SHUFFLE, S_CONN, S_TAU_R, S_TAU_D, S_PHI0, S_PHI1, S_WEIGHT, S_SPIKE, S_PAD, S_GAMMA_N, S_GAMMA_U, S_MULT, \
    S_SPONT, S_SP_TR, S_SP_TD, S_SP_T0, S_SP_W, S_GP, S_IID, S_GAMMA_B, S_TARGET = range(22)

GAMMA_TRIES = 16
GP_HALF_WIDTHS = 4          # filter support = +- 4 lengthscales

DEFAULTS = dict(N=300, T=900, H=10, trials=1000, nreps=1, connection_prob=0.05, powers=(45.0, 55.0, 65.0), min_latency=160.0,
                gamma_beta=15.0, sigma=6e-4, frac_strongly_connected=0.2, strong_weight_lower=20.0, strong_weight_upper=40.0,
                weak_exp_mean=4.0, min_weight=9.0, phi_0_lower=0.2, phi_0_upper=0.25, phi_1_lower=10.0, phi_1_upper=15.0,
                mult_noise_log_var=0.01, tau_r_min=25.0, tau_r_max=60.0, tau_delta_min=75.0, tau_delta_max=250.0, gp_scale=4e-3,
                gp_lengthscale=50.0, spont_prob=0.05, max_power_min_spike_rate=0.4, seed=0, design="blockwise")


def bits64(seed, stream, idx):
    """uint64 draw number `idx` of `stream`: key = (seed_lo, seed_hi ^ (0x5EED0000 + stream)), counter = (idx_lo, idx_hi)"""
    idx = np.asarray(idx, dtype=np.uint64)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32(((seed >> 32) & 0xFFFFFFFF) ^ (0x5EED0000 + stream))
    x0, x1 = threefry2x32(k0, k1, (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32))
    return (x0.astype(np.uint64) << np.uint64(32)) | x1.astype(np.uint64)


def uniform(seed, stream, idx):
    return (bits64(seed, stream, idx) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def normal(seed, stream, idx):
    """Box-Muller on draws 2 idx, 2 idx + 1"""
    idx = np.asarray(idx, dtype=np.uint64)
    u1 = uniform(seed, stream, 2 * idx)
    u2 = uniform(seed, stream, 2 * idx + np.uint64(1))
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def random_order(seed, stream, base, n):
    """permutation of range(n): positions sorted by (top 50 random bits, index)"""
    key = (bits64(seed, stream, base + np.arange(n, dtype=np.uint64)) >> np.uint64(14) << np.uint64(14)) | np.arange(n, dtype=np.uint64)
    return np.argsort(key, kind="stable").astype(np.int32)


def gamma(seed, idx, shape, scale):
    """Marsaglia-Tsang (2000) with a fixed budget of tries per variate; shape < 1 by the U^(1/a) boost"""
    idx = np.asarray(idx, dtype=np.uint64)
    shape = np.asarray(shape, dtype=np.float64)
    boost = shape < 1.0
    a = np.where(boost, shape + 1.0, shape)
    d = a - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    out = d.copy()                                   # value if every try is rejected (probability ~1e-20)
    done = np.zeros(idx.shape, dtype=bool)
    for j in range(GAMMA_TRIES):
        x = normal(seed, S_GAMMA_N, idx * np.uint64(GAMMA_TRIES) + np.uint64(j))
        u = uniform(seed, S_GAMMA_U, idx * np.uint64(GAMMA_TRIES) + np.uint64(j))
        v = 1.0 + c * x
        v3 = v * v * v
        with np.errstate(all="ignore"):
            ok = (v > 0.0) & (np.log(1.0 - u) < 0.5 * x * x + d - d * v3 + d * np.log(np.where(v > 0, v3, 1.0)))
        take = ok & ~done
        out = np.where(take, d * v3, out)
        done |= ok
    ub = uniform(seed, S_GAMMA_B, idx)
    out = np.where(boost, out * np.power(1.0 - ub, 1.0 / np.where(boost, shape, 1.0)), out)
    return out * scale


def geom_sum(tau, m):
    """sum_{j=0}^{m-1} exp(-j / tau)"""
    return (1.0 - np.exp(-m / tau)) / (1.0 - np.exp(-1.0 / tau))


def gp_filter(lengthscale):
    J = int(GP_HALF_WIDTHS * lengthscale)
    j = np.arange(-J, J + 1, dtype=np.float64)
    h = np.exp(-(j * j) / (lengthscale * lengthscale))
    return h / np.sqrt(np.sum(h * h)), J


def simulate(**kw):
    o = dict(DEFAULTS)
    unknown = set(kw) - set(o)
    if unknown:
        raise TypeError("unexpected argument %s" % sorted(unknown)[0])
    o.update(kw)
    N, T, H, K, nreps, seed = int(o["N"]), int(o["T"]), int(o["H"]), int(o["trials"]), int(o["nreps"]), int(o["seed"])
    powers = np.sort(np.asarray(o["powers"], dtype=np.float64))[::-1]            # highest power first (:49)
    P = len(powers)
    nholo = -(-N // H)
    tpr = P * nholo * nreps                                                        # trials per block
    R = -(-K // tpr)

    # ---- design
    stim = np.zeros((N, K))
    trial_power = np.zeros(K)
    trial_cells = -np.ones((K, H), dtype=np.int32)
    if o["design"] == "blockwise":
        orders = np.stack([random_order(seed, S_ORDER, np.uint64(r) * np.uint64(N), N) for r in range(R)])
        reorder = random_order(seed, S_SHUFFLE, np.uint64(0), K)
        for k in range(K):
            t = int(reorder[k])
            r, rem = divmod(t, tpr)
            pi, h = rem // (nholo * nreps), (rem // nreps) % nholo
            cells = orders[r, h * H:(h + 1) * H]
            stim[cells, k] = powers[pi]
            trial_power[k] = powers[pi]
            trial_cells[k, :len(cells)] = cells
    else:
        # 'random' (:65-77): a shuffled equal split of the powers (in the caller's order) over the trials, H targets per trial
        # drawn WITH replacement (a repeated cell collapses, so a trial can have fewer than H distinct targets)
        plist = np.asarray(o["powers"], dtype=np.float64)
        split = np.concatenate([np.full(len(a), p) for p, a in zip(plist, np.array_split(np.ones(K), len(plist)))])
        reorder = random_order(seed, S_SHUFFLE, np.uint64(0), K)
        for k in range(K):
            tars = np.floor(uniform(seed, S_TARGET, np.uint64(k) * np.uint64(H) + np.arange(H, dtype=np.uint64)) * N).astype(np.int32)
            trial_power[k] = split[reorder[k]]
            stim[tars, k] = trial_power[k]
            for i, n in enumerate(tars):
                trial_cells[k, i] = n if n not in tars[:i] else -1
    # ---- per-neuron parameters
    n_idx = np.arange(N, dtype=np.uint64)
    tau_r = o["tau_r_min"] + (o["tau_r_max"] - o["tau_r_min"]) * uniform(seed, S_TAU_R, n_idx)
    tau_d = tau_r + o["tau_delta_min"] + (o["tau_delta_max"] - o["tau_delta_min"]) * uniform(seed, S_TAU_D, n_idx)
    phi_0 = o["phi_0_lower"] + (o["phi_0_upper"] - o["phi_0_lower"]) * uniform(seed, S_PHI0, n_idx)
    phi_1 = o["phi_1_lower"] + (o["phi_1_upper"] - o["phi_1_lower"]) * uniform(seed, S_PHI1, n_idx)
    # kernel area: trapz over [0, T) of exp(-j/tau_d) - exp(-j/tau_r)  (+ 1e-5)     (:285-289); ke[0] = 0
    area = (geom_sum(tau_d, T) - geom_sum(tau_r, T)) - 0.5 * (np.exp(-(T - 1) / tau_d) - np.exp(-(T - 1) / tau_r)) + 1e-5
    conn = random_order(seed, S_CONN, np.uint64(0), N)
    rank = np.empty(N, dtype=np.int64)
    rank[conn] = np.arange(N)
    n_conn = int(o["connection_prob"] * N)
    n_strong = int(math.ceil(o["frac_strongly_connected"] * n_conn))
    uw = uniform(seed, S_WEIGHT, n_idx)
    weights = np.where(rank < n_strong, o["strong_weight_lower"] + (o["strong_weight_upper"] - o["strong_weight_lower"]) * uw,
                       np.where(rank < n_conn, o["min_weight"] - o["weak_exp_mean"] * np.log(1.0 - uw), 0.0))
    wlo = np.min(weights[rank < n_conn]) if n_conn else 0.0
    whi = np.max(weights[rank < n_conn]) if n_conn else 0.0

    # ---- spikes, padded to the minimum rate at maximal power
    nn, kk = np.nonzero(stim)
    e_idx = nn.astype(np.uint64) * np.uint64(K) + kk.astype(np.uint64)
    frate = 1.0 / (1.0 + np.exp(-(phi_0[nn] * stim[nn, kk] - phi_1[nn])))
    spks = np.zeros((N, K))
    spks[nn, kk] = (uniform(seed, S_SPIKE, e_idx) <= frate).astype(np.float64)
    pmax = powers[0]
    for n in range(N):
        locs = np.flatnonzero(stim[n] == pmax)
        if len(locs) == 0:
            continue
        diff = o["max_power_min_spike_rate"] - np.mean(spks[n, locs])
        if diff > 0:
            zero = locs[spks[n, locs] == 0]
            req = min(int(math.ceil(diff * len(locs))), len(zero))
            key = (bits64(seed, S_PAD, np.uint64(n) * np.uint64(K) + zero.astype(np.uint64)) >> np.uint64(14) << np.uint64(14)) \
                | zero.astype(np.uint64)
            spks[n, zero[np.argsort(key, kind="stable")[:req]]] = 1.0

    # ---- latencies, gains, evoked PSCs
    nn, kk = np.nonzero(spks)
    e_idx = nn.astype(np.uint64) * np.uint64(K) + kk.astype(np.uint64)
    spk_times = np.zeros((N, K))
    spk_times[nn, kk] = o["min_latency"] + gamma(seed, e_idx, 1e4 / stim[nn, kk] ** 2, o["gamma_beta"])
    mult = np.exp(o["mult_noise_log_var"] * normal(seed, S_MULT, e_idx))
    tgrid = np.arange(T, dtype=np.float64)
    psc = np.zeros((K, T))
    for n, k, m in zip(nn, kk, mult):
        s = int(spk_times[n, k])
        if weights[n] == 0.0 or s >= T:
            continue
        j = tgrid - s
        ke = np.where(j >= 0, np.exp(-j / tau_d[n]) - np.exp(-j / tau_r[n]), 0.0) / area[n]
        tot = (geom_sum(tau_d[n], T - s) - geom_sum(tau_r[n], T - s)) / area[n]
        psc[k] += ke / (tot + 1e-5) * (m * weights[n])

    # ---- spontaneous PSCs
    k_idx = np.arange(K, dtype=np.uint64)
    spont = uniform(seed, S_SPONT, k_idx) <= o["spont_prob"]
    sp_tr = o["tau_r_min"] + (o["tau_r_max"] - o["tau_r_min"]) * uniform(seed, S_SP_TR, k_idx)
    sp_td = sp_tr + o["tau_delta_min"] + (o["tau_delta_max"] - o["tau_delta_min"]) * uniform(seed, S_SP_TD, k_idx)
    sp_t0 = 1 + np.floor(uniform(seed, S_SP_T0, k_idx) * (T - 1))                  # randint(1, T)
    sp_w = wlo + (whi - wlo) * uniform(seed, S_SP_W, k_idx)
    spont_pscs = np.zeros((K, T))
    for k in np.flatnonzero(spont):
        j = tgrid - sp_t0[k]
        kern = np.where(j > 0, np.exp(-j / sp_td[k]) - np.exp(-j / sp_tr[k]), 0.0)
        spont_pscs[k] = sp_w[k] * kern / (np.sum(kern) - 0.5 * (kern[0] + kern[-1]) + 1e-5)

    # ---- correlated + iid noise
    h, J = gp_filter(o["gp_lengthscale"])
    W = T + 2 * J
    z = normal(seed, S_GP, (k_idx[:, None] * np.uint64(W) + np.arange(W, dtype=np.uint64)[None, :]))
    gp = np.zeros((K, T))
    for q in range(2 * J + 1):
        gp += h[q] * z[:, q:q + T]
    gp *= o["gp_scale"]
    iid = o["sigma"] * normal(seed, S_IID, k_idx[:, None] * np.uint64(T) + np.arange(T, dtype=np.uint64)[None, :])
    psc = psc + spont_pscs + gp + iid
    return {"weights": weights, "phi_0": phi_0, "phi_1": phi_1, "sigma": o["sigma"], "stim_matrix": stim, "psc": psc,
            "gp_noise": gp, "spks": spks, "spk_times": spk_times, "spont_pscs": spont_pscs, "I": trial_power,
            "tau_r": tau_r, "tau_d": tau_d, "trial_cells": trial_cells}
