"""TEST INFRASTRUCTURE ONLY -- float64 CPU restatement of the reference's CAVIaR path.

Follows ``circuitmap/optimise/caviar.py`` and ``circuitmap/optimise/pava.py`` of the reference
operation for operation (citations are file:line relative to /root/reference).  Third-party
arithmetic the reference reaches through jax (``ndtr``/``ndtri``, ``linalg.inv``, ``sigmoid``,
the threefry PRNG) is substituted by SciPy / LAPACK / ``oracle/threefry.py``.

PARITY PIN: the reference holds no tests, golden vectors or fixtures for this path (SURVEY.md
section 4) and cannot be imported here (jax absent).  This oracle is pinned by
``tests/golden/*.npz`` -- outputs of the reference's OWN source files executed in the build
container under a NumPy-backed stand-in for the small jax surface they use
(``tests/golden/make_golden.py``).  That pins the control flow and every quirk of the reference
source; the third-party jax numerics underneath (ndtr/ndtri/inv/PRNG bit streams) remain
unpinned against a live jax.

Two formulations of each phase are kept and cross-checked in tests:
  * ``literal=True``  -- dense O(N^2 K) leave-one-out sums, per-element Newton sums, LU inverse.
  * ``literal=False`` -- same maths with the running residual and per-power sufficient
    statistics (what makes cfg2/cfg3-size parity checks finish in seconds).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
The product path (circuitmap_amd) never does.
"""
import numpy as np
from scipy.special import ndtr, ndtri
from scipy.stats import sem

from . import threefry

_trapz = getattr(np, "trapezoid", None) or np.trapz


def sigmoid(x):
    """jax.nn.sigmoid == scipy expit (logistic), f64."""
    from scipy.special import expit
    return expit(x)


# --------------------------------------------------------------------------------------------
# pava.py
# --------------------------------------------------------------------------------------------
def isotonic_regression(y):
    """pava.py:9-88 with gamma=1 (the only value the hot path uses): pool-adjacent-violators,
    non-decreasing fit, merge on STRICT '>' (pava.py:42), unit weights."""
    y = np.asarray(y, dtype=np.float64)
    T = len(y)
    v = np.zeros(T)
    w = np.zeros(T)
    l = np.zeros(T, dtype=np.int64)
    v[0], w[0], l[0] = y[0], 1.0, 1
    i = 0
    for t in range(1, T):                      # pava.py:23-37
        i += 1
        v[i], w[i], l[i] = y[t], 1.0, 1
        while i > 0 and (v[i - 1] / w[i - 1]) > (v[i] / w[i]):   # pava.py:39-43
            i -= 1                             # pava.py:45-56
            v[i] += v[i + 1]
            w[i] += w[i + 1]
            l[i] += l[i + 1]
            v[i + 1] = w[i + 1] = 0.0
            l[i + 1] = 0
    out = np.zeros(T)                          # pava.py:63-80
    t = 0
    for j in range(T):
        if l[j] > 0:
            out[t:t + l[j]] = v[j] / w[j]
            t += l[j]
    return out


def pava_last(y):
    """Closed form of ``isotonic_regression(y)[-1]``: max_j mean(y[j:]) (checked in tests)."""
    y = np.asarray(y, dtype=np.float64)
    best = -np.inf
    s = 0.0
    for j in range(len(y) - 1, -1, -1):
        s += y[j]
        m = s / (len(y) - j)
        if m > best:
            best = m
    return best


# --------------------------------------------------------------------------------------------
# prologue  (caviar.py:28-51)
# --------------------------------------------------------------------------------------------
def prologue(y_psc, I, y_xcorr_thresh=1e-2):
    y_psc = np.asarray(y_psc, dtype=np.float64)
    y = _trapz(y_psc, axis=-1)                                   # :28
    lam_mask = np.sum(y_psc * y_psc, axis=-1) > y_xcorr_thresh   # :30 (valid self-correlation)
    I = np.asarray(I).astype(float)                              # :35
    lam = np.zeros_like(I)                                       # :32
    lam[I > 0] = 0.95                                            # :33
    lam = lam * lam_mask                                         # :34
    powers = np.unique(I)[1:]                                    # :42
    return y, lam_mask, lam, I, powers


# --------------------------------------------------------------------------------------------
# block_update_mu  (caviar.py:166-172)
# --------------------------------------------------------------------------------------------
def block_update_mu(y, lam, shape, rate, mu_prior, beta_prior, literal=True):
    N = lam.shape[0]
    sig = shape / rate
    if literal:
        L = lam @ lam.T
        D = np.diag(np.sum(lam * (1 - lam), axis=-1))
        posterior_cov = np.linalg.inv(sig * (D + L) + 1 / (beta_prior ** 2) * np.eye(N))
        posterior_mean = posterior_cov @ (sig * np.sum(y * lam, axis=1) + 1 / (beta_prior ** 2) * mu_prior)
        return posterior_mean, np.diag(posterior_cov).copy()
    # rows with lam == 0 decouple exactly: mean = mu_prior, var = beta_prior^2 (SURVEY 3.2 note)
    act = np.flatnonzero(np.any(lam != 0, axis=1))
    mean = np.array(mu_prior, dtype=np.float64, copy=True)
    var = np.asarray(beta_prior, dtype=np.float64) ** 2 * np.ones(N)
    if len(act):
        la = lam[act]
        A = sig * (la @ la.T + np.diag(np.sum(la * (1 - la), axis=-1))) + np.diag(1 / beta_prior[act] ** 2)
        b = sig * (la @ y) + mu_prior[act] / beta_prior[act] ** 2
        c = np.linalg.cholesky(A)
        ci = np.linalg.solve(c, np.eye(len(act)))
        mean[act] = ci.T @ (ci @ b)
        var[act] = np.sum(ci * ci, axis=0)
    return mean, var


# --------------------------------------------------------------------------------------------
# update_lam  (caviar.py:190-236)
# --------------------------------------------------------------------------------------------
def sample_phi_mc(phi_n, phi_cov_n, u):
    """caviar.py:211-212.  NB ``sdev`` is the VARIANCE diagonal used as a std-dev (quirk 2)."""
    mean, sdev = phi_n, np.diag(phi_cov_n)
    with np.errstate(all="ignore"):
        c = ndtr(-mean / sdev)
        return ndtri(c + u * (1 - c)) * sdev + mean


def mc_logit(Irow, mc):
    """caviar.py:215,233-236: mean over samples of log(f/(1-f)), f = sigmoid(phi0*I - phi1)."""
    with np.errstate(all="ignore"):
        fn = sigmoid(mc[:, 0:1] * Irow[None, :] - mc[:, 1:2])
        return np.mean(np.log(fn / (1 - fn)), axis=0)


def eval_spike_rates(stimv, lamv, powers):
    """caviar.py:174-186: mean of lamv over trials with stimv == power (exact float equality);
    an empty group gives 0 / 1e-4 = 0."""
    out = np.zeros(len(powers))
    for p, power in enumerate(powers):
        m = stimv == power
        cnt = np.sum(m)
        out[p] = np.sum(lamv[m]) / (cnt + 1e-4 * (cnt == 0))
    return out


def update_lam(y, I, mu, beta, lam, shape, rate, phi, phi_cov, lam_mask, stream, it,
               num_mc_samples, powers, minimum_spike_count, minimax_spk_prob, delay_spont_est,
               literal=True, trace=None):
    N, K = I.shape
    order = np.asarray(stream.update_order(it, N))        # :196
    sig = shape / rate                                    # :197
    lam = lam.copy()
    gates = np.ones(N)
    if not literal:
        return _update_lam_fast(y, I, mu, beta, lam, sig, phi, phi_cov, lam_mask, stream, it, num_mc_samples, powers,
                                minimum_spike_count, minimax_spk_prob, delay_spont_est, order, gates)
    mu_c = mu.copy()                                      # in-sweep mu edits are discarded (:225,231)
    for m in range(N):                                    # :229
        n = int(order[m])                                 # :203
        mask = np.arange(N) != n                          # :204
        loo = np.sum(sig * mu_c[mask][:, None] * lam[mask], 0)
        arg = -2 * sig * y * mu_c[n] + 2 * mu_c[n] * loo + sig * (mu_c[n] ** 2 + beta[n] ** 2)   # :205-206
        u = stream.uniforms(it, m, num_mc_samples)        # :209-210
        mc = sample_phi_mc(phi[n], phi_cov[n], u)         # :211-212
        mcE = mc_logit(I[n], mc)                          # :215
        with np.errstate(all="ignore"):
            est = lam_mask * (I[n] > 0) * sigmoid(mcE - 0.5 * arg)       # :216
        srates = eval_spike_rates(I[n], est, powers)      # :219
        iso_last = isotonic_regression(srates)[-1]
        pava = float(iso_last >= minimax_spk_prob) * float(np.sum(est) >= minimum_spike_count)   # :220
        pava = pava * (it > delay_spont_est) + 1.0 * (it <= delay_spont_est)                        # :221
        lam[n] = est * pava                               # :224
        mu_c[n] = mu_c[n] * (pava == 1.0)                 # :225 (no numerical effect; kept literal)
        gates[n] = pava
        if trace is not None:
            trace.setdefault("srates", {})[(it, n)] = srates
    return lam, gates


def _update_lam_fast(y, I, mu, beta, lam, sig, phi, phi_cov, lam_mask, stream, it, S, powers, minimum_spike_count,
                     minimax_spk_prob, delay_spont_est, order, gates):
    """Same sweep on the targeted trials only, with the running residual R = sum_m mu_m lam_m
    (leave-one-out = R - mu_n lam_n) and the Monte-Carlo logit evaluated once per distinct power."""
    N, K = I.shape
    R = mu @ lam
    cnt = np.array([[np.sum(I[n] == p) for p in powers] for n in range(N)]) if len(powers) else np.zeros((N, 0))
    for m in range(N):
        n = int(order[m])
        idx = np.flatnonzero(I[n] > 0)
        Iv = I[n, idx]
        u = stream.uniforms(it, m, S)
        mc = sample_phi_mc(phi[n], phi_cov[n], u)
        vals, inv = np.unique(Iv, return_inverse=True)
        mcE = mc_logit(vals, mc)[inv] if len(idx) else np.zeros(0)
        loo = sig * (R[idx] - mu[n] * lam[n, idx])
        arg = -2 * sig * y[idx] * mu[n] + 2 * mu[n] * loo + sig * (mu[n] ** 2 + beta[n] ** 2)
        with np.errstate(all="ignore"):
            est = lam_mask[idx] * sigmoid(mcE - 0.5 * arg)
        srates = np.array([np.sum(est[Iv == p]) / (c + 1e-4 * (c == 0)) for p, c in zip(powers, cnt[n])])
        iso_last = pava_last(srates) if len(srates) else 0.0
        pava = float(iso_last >= minimax_spk_prob) * float(np.sum(est) >= minimum_spike_count)
        pava = pava * (it > delay_spont_est) + 1.0 * (it <= delay_spont_est)
        new = est * pava
        R[idx] += mu[n] * (new - lam[n, idx])
        lam[n, idx] = new
        gates[n] = pava
    return lam, gates


# --------------------------------------------------------------------------------------------
# update_sigma  (caviar.py:238-244)
# --------------------------------------------------------------------------------------------
def update_sigma(y, mu, beta, lam, shape_prior, rate_prior):
    K = y.shape[0]
    shape = shape_prior + K / 2
    rate = rate_prior + 0.5 * (np.sum(np.square(y - np.sum(mu[:, None] * lam, 0)))
                               - np.sum(np.square(mu[:, None] * lam))
                               + np.sum((mu ** 2 + beta ** 2)[:, None] * lam))
    return shape, rate


# --------------------------------------------------------------------------------------------
# update_phi / laplace_approx  (caviar.py:246-316)
# --------------------------------------------------------------------------------------------
def _negloglik_with_barrier(y, phi, phi_prior, prec, I, t):
    """caviar.py:312-316 -- nan_to_num is applied PER ELEMENT before the sum (quirk 11)."""
    with np.errstate(all="ignore"):
        lam = sigmoid(phi[0] * I - phi[1])
        ll = np.nan_to_num(y * np.log(lam) + (1 - y) * np.log(1 - lam))
        d = phi - phi_prior
        return -np.sum(ll) - np.sum(np.log(phi)) / t + 0.5 * d @ prec @ d


def laplace_approx_one(y, phi_prior, phi_cov, I, t=1e1, backtrack_alpha=0.25, backtrack_beta=0.5,
                       max_backtrack_iters=40, newton_steps=10):
    """caviar.py:253-308 for one neuron: y = lam[n,:], I = I[n,:] (all K trials, I == 0 included)."""
    phi = np.array(phi_prior, dtype=np.float64, copy=True)     # :305 (no warm start, quirk 3)
    prior_prec = np.linalg.inv(phi_cov)                        # :306
    cov = np.zeros((2, 2))
    with np.errstate(all="ignore"):
        for _ in range(newton_steps):                          # :308
            f = sigmoid(phi[0] * I - phi[1])                   # :275
            j1 = -np.sum(I * (y - f))
            j2 = np.sum(y - f)
            J = np.array([j1, j2]) + prior_prec @ (phi - phi_prior) - 1 / (t * phi)    # :278-280
            h11 = np.sum(I ** 2 * f * (1 - f))
            h12 = -np.sum(I * f * (1 - f))
            h22 = np.sum(f * (1 - f))
            H = np.array([[h11, h12], [h12, h22]]) + prior_prec + np.diag(1 / (t * phi ** 2))   # :283-287
            H_inv = np.linalg.inv(H)
            v = -H_inv @ J
            cov = H_inv                                        # :295,302: covariance BEFORE the update
            step = 1.0
            f0 = _negloglik_with_barrier(y, phi, phi_prior, prior_prec, I, t)
            lhs = _negloglik_with_barrier(y, phi + step * v, phi_prior, prior_prec, I, t)
            rhs = f0 + backtrack_alpha * step * J @ v
            bt = 0
            while bt < max_backtrack_iters and (np.isnan(lhs) or lhs > rhs):          # :259-261
                bt += 1
                step *= backtrack_beta
                lhs = _negloglik_with_barrier(y, phi + step * v, phi_prior, prior_prec, I, t)
                rhs = f0 + backtrack_alpha * step * J @ v
            phi = phi + step * v                               # :301
    return phi, cov


def phi_suffstats(lam_row, I_row, powers):
    """Per-power sufficient statistics of one row.  Group 0 is I == 0 (where lam == 0 by the
    support invariant); groups 1..P the distinct powers.  For each group: count, #(y == 0),
    #(y == 1), and sum / count of the strictly interior y."""
    vals = np.concatenate([[0.0], powers])
    st = np.zeros((len(vals), 5))
    for g, v in enumerate(vals):
        yy = lam_row[I_row == v]
        mid = yy[(yy > 0) & (yy < 1)]
        st[g] = (len(yy), np.sum(yy == 0), np.sum(yy == 1), np.sum(mid), len(mid))
    return vals, st


def laplace_approx_one_suff(vals, st, phi_prior, phi_cov, t=1e1, backtrack_alpha=0.25, backtrack_beta=0.5,
                            max_backtrack_iters=40, newton_steps=10):
    """Same Newton/backtracking iteration as `laplace_approx_one` on the sufficient statistics."""
    cnt, n0, n1, smid, nmid = st.T
    sy = n1 + smid

    def nll(phi):
        with np.errstate(all="ignore"):
            s = sigmoid(phi[0] * vals - phi[1])
            ls, l1s = np.log(s), np.log(1 - s)
            tot = 0.0
            for g in range(len(vals)):
                if n0[g]:
                    tot += n0[g] * np.nan_to_num(0.0 * ls[g] + 1.0 * l1s[g])
                if n1[g]:
                    tot += n1[g] * np.nan_to_num(1.0 * ls[g] + 0.0 * l1s[g])
                if nmid[g]:
                    if np.isfinite(ls[g]) and np.isfinite(l1s[g]):
                        tot += smid[g] * ls[g] + (nmid[g] - smid[g]) * l1s[g]
                    else:   # every interior element saturates the same way: -inf -> -DBL_MAX each
                        tot += nmid[g] * np.nan_to_num(0.5 * ls[g] + 0.5 * l1s[g])
            d = phi - phi_prior
            return -tot - np.sum(np.log(phi)) / t + 0.5 * d @ prior_prec @ d

    phi = np.array(phi_prior, dtype=np.float64, copy=True)
    prior_prec = np.linalg.inv(phi_cov)
    cov = np.zeros((2, 2))
    with np.errstate(all="ignore"):
        for _ in range(newton_steps):
            f = sigmoid(phi[0] * vals - phi[1])
            j1 = -np.sum(vals * (sy - cnt * f))
            j2 = np.sum(sy - cnt * f)
            J = np.array([j1, j2]) + prior_prec @ (phi - phi_prior) - 1 / (t * phi)
            ff = cnt * f * (1 - f)
            H = np.array([[np.sum(vals ** 2 * ff), -np.sum(vals * ff)],
                          [-np.sum(vals * ff), np.sum(ff)]]) + prior_prec + np.diag(1 / (t * phi ** 2))
            H_inv = np.linalg.inv(H)
            v = -H_inv @ J
            cov = H_inv
            step = 1.0
            f0 = nll(phi)
            lhs = nll(phi + step * v)
            rhs = f0 + backtrack_alpha * step * J @ v
            bt = 0
            while bt < max_backtrack_iters and (np.isnan(lhs) or lhs > rhs):
                bt += 1
                step *= backtrack_beta
                lhs = nll(phi + step * v)
                rhs = f0 + backtrack_alpha * step * J @ v
            phi = phi + step * v
    return phi, cov


_PHI_MEMO = {}


def update_phi(lam, I, phi_prior, phi_cov_prior, powers=None, literal=True):
    """caviar.py:246-251,310: independent Laplace approximations for all N cells."""
    N = lam.shape[0]
    phi = np.zeros((N, 2))
    phi_cov = np.zeros((N, 2, 2))
    for n in range(N):
        if literal:
            phi[n], phi_cov[n] = laplace_approx_one(lam[n], phi_prior[n], phi_cov_prior[n], I[n])
        else:
            vals, st = phi_suffstats(lam[n], I[n], powers)
            key = (vals.tobytes(), st.tobytes(), phi_prior[n].tobytes(), phi_cov_prior[n].tobytes())
            if key not in _PHI_MEMO:          # rows that stay all-zero repeat the same problem every iteration
                if len(_PHI_MEMO) > 200000:
                    _PHI_MEMO.clear()
                _PHI_MEMO[key] = laplace_approx_one_suff(vals, st, phi_prior[n], phi_cov_prior[n])
            phi[n], phi_cov[n] = _PHI_MEMO[key]
    return phi, phi_cov


# --------------------------------------------------------------------------------------------
# estimate_spont_act_soft_thresh  (caviar.py:147-163)
# --------------------------------------------------------------------------------------------
def estimate_spont_act_soft_thresh(y, mu, lam, it, max_iters, z, pen, mask, scale_factor, tol,
                                   spont_orthogonality):
    """The while-loop counter is seeded with the OUTER iteration index and err with sum(y)
    (caviar.py:86): at most ``max_iters - it`` penalty steps; none once it >= max_iters (quirk 4)."""
    err = np.sum(y)
    while it < max_iters and err > tol:                                  # :147-149
        resid = y - lam.T @ mu                                            # :153
        z = np.where(resid < pen, 0.0, resid - pen)                       # :154
        z = np.where(z < 0.0, 0.0, z)                                     # :155
        z = np.where(np.any(lam >= spont_orthogonality, axis=0), 0.0, z)  # :156
        z = z * mask                                                      # :157
        err = np.sum(np.square(resid - z)) / (np.sum(np.square(y)) + 1e-5)    # :158
        it += 1
        pen *= scale_factor
    return z


# --------------------------------------------------------------------------------------------
# reconnect_spont_cells  (caviar.py:102-144)
# --------------------------------------------------------------------------------------------
def reconnect_spont_cells(y, stim_matrix, lam, mu, beta, z, minimax_spk_prob=0.3, minimum_spike_count=3):
    mu, beta, lam, z = mu.copy(), beta.copy(), lam.copy(), np.array(z, copy=True)
    disc_cells = np.where(mu == 0.0)[0]                                   # :103 exact zero test
    powers = np.unique(stim_matrix)[1:]                                   # :104
    while len(disc_cells) > 0:                                            # :109
        if len(np.where(z)[0]) > minimum_spike_count:                     # :111
            stim_locs = [np.where(z[np.where(stim_matrix[n])[0]])[0] for n in disc_cells]    # :112-114
            focus_indx = int(np.argmax([len(sl) for sl in stim_locs]))    # :117
            focus = disc_cells[focus_indx]
            srates = np.zeros_like(powers)                                # :121
            spike_count = 0
            for i, p in enumerate(powers):
                z_locs = np.where(stim_matrix[focus] == p)[0]
                if len(z_locs) > 0:
                    srates[i] = np.mean(z[z_locs] != 0)
                    spike_count += np.sum(z[z_locs] != 0)
            pava = isotonic_regression(srates)[-1]                        # :128
            if pava >= minimax_spk_prob and spike_count >= minimum_spike_count:      # :130
                z_locs = np.intersect1d(np.where(stim_matrix[focus])[0], np.where(z)[0])
                mu[focus] = np.mean(z[z_locs])                            # :134
                with np.errstate(all="ignore"):
                    beta[focus] = sem(z[z_locs])                          # :135 (ddof=1; NaN for one event)
                lam[focus, z_locs] = 1.0                                  # :136
                z[z_locs] = 0.0                                           # :137
            disc_cells = np.delete(disc_cells, focus_indx)                # :139
        else:
            break                                                         # :141
    return mu, beta, lam, z


# --------------------------------------------------------------------------------------------
# driver  (caviar.py:20-100)
# --------------------------------------------------------------------------------------------
def caviar(y_psc, I, mu_prior, beta_prior, shape_prior, rate_prior, phi_prior, phi_cov_prior,
           iters=50, num_mc_samples=100, seed=0, y_xcorr_thresh=1e-2, minimum_spike_count=3,
           delay_spont_est=1, msrmp=0.3, scale_factor=0.75, penalty=5e0, save_histories=False,
           max_backtrack_iters=20, tol=0.05, spont_orthogonality=0.1, fn_scan=True,
           stream=None, literal=True, on_iteration=None):
    """Returns the reference's 17-tuple (caviar.py:100).  ``stream`` overrides the seeded threefry
    stream with explicit RNG streams (oracle/threefry.ExplicitStream)."""
    y, lam_mask, lam, I, powers = prologue(y_psc, I, y_xcorr_thresh)
    spont_rate = 0.0
    mu_prior = np.asarray(mu_prior, dtype=np.float64)
    beta_prior = np.asarray(beta_prior, dtype=np.float64)
    phi_prior = np.asarray(phi_prior, dtype=np.float64)
    phi_cov_prior = np.asarray(phi_cov_prior, dtype=np.float64)
    N, K = mu_prior.shape[0], y.shape[0]
    mu, beta = mu_prior.copy(), beta_prior.copy()
    shape, rate = float(shape_prior), float(rate_prior)
    phi, phi_cov = phi_prior.copy(), phi_cov_prior.copy()
    z = np.zeros(K)
    receptive_fields = None
    if save_histories:
        hist = [np.zeros((iters, N)), np.zeros((iters, N)), np.zeros((iters, N, K)), np.zeros((iters, K)),
                np.zeros((iters, K)), np.zeros((iters, N, 2)), np.zeros((iters, N, 2, 2)), np.zeros((iters, K))]
    else:
        hist = [None] * 8
    if stream is None:
        stream = threefry.JaxLikeStream(seed)

    for it in range(iters):                                                   # :79
        mu, beta = block_update_mu(y, lam, shape, rate, mu_prior, beta_prior, literal)          # :80
        lam, gates = update_lam(y, I, mu, beta, lam, shape, rate, phi, phi_cov, lam_mask, stream, it,
                                num_mc_samples, powers, minimum_spike_count, msrmp + spont_rate,
                                delay_spont_est, literal)                     # :81-83
        shape, rate = update_sigma(y, mu, beta, lam, shape_prior, rate_prior)                   # :84
        phi, phi_cov = update_phi(lam, I, phi_prior, phi_cov_prior, powers, literal)            # :85
        stream.end_iteration(it)
        z = estimate_spont_act_soft_thresh(y, mu, lam, it, max_backtrack_iters, z, penalty, lam_mask,
                                           scale_factor, tol, spont_orthogonality)              # :86-87
        spont_rate = np.mean(z != 0.0)                                        # :88
        if save_histories:                                                    # :90-92
            for h, pa in zip(hist, [mu, beta, lam, shape, rate, phi, phi_cov, z]):
                h[it] = pa
        if on_iteration is not None:
            on_iteration(it, dict(mu=mu, beta=beta, lam=lam, shape=shape, rate=rate, phi=phi,
                                  phi_cov=phi_cov, z=z, spont_rate=spont_rate, gates=gates))

    if fn_scan:                                                               # :94-98
        mu, beta, lam, z = reconnect_spont_cells(y, I, lam, mu, beta, z, minimax_spk_prob=msrmp,
                                                 minimum_spike_count=minimum_spike_count)
        phi, phi_cov = update_phi(lam, I, phi_prior, phi_cov_prior, powers, literal)

    return (mu, beta, lam, shape, rate, phi, phi_cov, z, receptive_fields, *hist)
