"""`caviar()` with the reference's signature and 17-tuple return, running on libcaviar_hip (MI355X).

Drop-in for ``circuitmap.optimise.caviar`` (reference circuitmap/optimise/caviar.py:20-23, return
value :100).  Host code only marshals arguments; every numerical phase runs in HIP kernels.
"""
import ctypes as C

import numpy as np

from .. import _lib

_OPTION_NAMES = ["iters", "num_mc_samples", "seed", "y_xcorr_thresh", "minimum_spike_count", "delay_spont_est", "msrmp",
                 "scale_factor", "penalty", "save_histories", "max_backtrack_iters", "tol", "spont_orthogonality", "fn_scan"]


def make_options(lib, iters=50, num_mc_samples=100, seed=0, y_xcorr_thresh=1e-2, minimum_spike_count=3,
                 delay_spont_est=1, msrmp=0.3, scale_factor=0.75, penalty=5e0, save_histories=False,
                 max_backtrack_iters=20, tol=0.05, spont_orthogonality=0.1, fn_scan=True, verbose=0, timing=0, interleave=True):
    o = _lib.CmOptions()
    lib.cm_options_default(C.byref(o))
    o.iters, o.num_mc_samples, o.seed = int(iters), int(num_mc_samples), int(seed)
    o.y_xcorr_thresh, o.minimum_spike_count = float(y_xcorr_thresh), float(minimum_spike_count)
    o.delay_spont_est, o.msrmp, o.scale_factor, o.penalty = int(delay_spont_est), float(msrmp), float(scale_factor), float(penalty)
    o.save_histories, o.max_backtrack_iters, o.tol = int(bool(save_histories)), int(max_backtrack_iters), float(tol)
    o.spont_orthogonality, o.fn_scan, o.verbose = float(spont_orthogonality), int(bool(fn_scan)), int(verbose)
    o.timing = int(timing)
    o.interleave = 0 if interleave else 1
    return o


class Problem:
    """A circuit resident in HBM: encoded stimulus + PSC integrals + CAVIaR state (C-ABI cm_problem)."""

    def __init__(self, ctx, y_psc, I, device_ptrs=None):
        self.ctx = ctx
        self.lib = ctx.lib
        self.handle = C.c_void_p()
        if device_ptrs is not None:
            N, K, T, psc_ptr, stim_ptr = device_ptrs
            kind = _lib.CM_MEM_DEVICE
            self._keep = None
        else:
            psc = _lib.f64(y_psc)
            stim = _lib.f64(I)                                    # caviar.py:35 (callers may pass ints); no copy if already float64
            if psc.ndim != 2 or stim.ndim != 2 or psc.shape[0] != stim.shape[1]:
                raise ValueError("y_psc must be K x T and I must be N x K (got %s, %s)" % (psc.shape, stim.shape))
            N, K, T = stim.shape[0], stim.shape[1], psc.shape[1]
            psc_ptr, stim_ptr = psc.ctypes.data, stim.ctypes.data
            kind = _lib.CM_MEM_HOST
            self._keep = (psc, stim)
        self.N, self.K, self.T = int(N), int(K), int(T)
        with ctx.lock:
            _lib.check(self.lib.cm_problem_create(ctx.handle, self.N, self.K, self.T, C.c_void_p(psc_ptr), C.c_void_p(stim_ptr),
                                                  kind, C.byref(self.handle)))
        self._keep = None

    @classmethod
    def create_batch(cls, ctx, circuits, compact=False):
        """cm_problem_create_batch: `circuits` = [(psc K x T, stim N x K), ...] host arrays or [("device", N, K, T, psc_ptr, stim_ptr), ...]
        (all of one kind) -> list of Problems; the prologue kernels run once for the group.  compact=True (host arrays): psc travels as
        float32 and the stimulus matrix as its non-zero (trial, neuron, power) entries (cm_problem_create_batch_ex: a quarter of the bytes over
        PCIe; bitwise the same circuits when the values are float32-representable, which recorded traces and laser powers are)."""
        n = len(circuits)
        if n == 0:
            return []
        if compact:
            if len(circuits[0]) != 2:
                raise ValueError("create_batch(compact=True) takes host arrays")
            keep, dims = [], []
            ins = (_lib.CmInputs * n)()
            for i, (psc, stim) in enumerate(circuits):
                psc, stim = np.asarray(psc), np.asarray(stim)
                if psc.ndim != 2 or stim.ndim != 2 or psc.shape[0] != stim.shape[1]:
                    raise ValueError("y_psc must be K x T and I must be N x K (got %s, %s)" % (psc.shape, stim.shape))
                ins[i] = _lib.compact_inputs(psc, stim, keep)
                dims.append((int(stim.shape[0]), int(stim.shape[1]), int(psc.shape[1])))
            i32 = C.c_int32 * n
            handles = (C.c_void_p * n)()
            with ctx.lock:
                _lib.check(ctx.lib.cm_problem_create_batch_ex(ctx.handle, n, i32(*[d[0] for d in dims]), i32(*[d[1] for d in dims]), i32(*[d[2] for d in dims]),
                                                              ins, _lib.CM_MEM_HOST, handles))
            out = []
            for d, h in zip(dims, handles):
                q = cls.__new__(cls)
                q.ctx, q.lib, q.handle, q._keep = ctx, ctx.lib, C.c_void_p(h), None
                q.N, q.K, q.T = d
                out.append(q)
            return out
        dev = len(circuits[0]) != 2
        keep, dims, ptrs = [], [], []
        for c in circuits:
            if (len(c) != 2) != dev:
                raise ValueError("create_batch: host and device circuits cannot be mixed")
            if dev:
                _, N, K, T, psc_ptr, stim_ptr = c
            else:
                psc, stim = _lib.f64(c[0]), _lib.f64(c[1])
                if psc.ndim != 2 or stim.ndim != 2 or psc.shape[0] != stim.shape[1]:
                    raise ValueError("y_psc must be K x T and I must be N x K (got %s, %s)" % (psc.shape, stim.shape))
                keep += [psc, stim]
                N, K, T = stim.shape[0], stim.shape[1], psc.shape[1]
                psc_ptr, stim_ptr = psc.ctypes.data, stim.ctypes.data
            dims.append((int(N), int(K), int(T)))
            ptrs.append((psc_ptr, stim_ptr))
        i32 = C.c_int32 * n
        vp = C.c_void_p * n
        handles = vp()
        with ctx.lock:
            _lib.check(ctx.lib.cm_problem_create_batch(ctx.handle, n, i32(*[d[0] for d in dims]), i32(*[d[1] for d in dims]), i32(*[d[2] for d in dims]),
                                                       vp(*[p[0] for p in ptrs]), vp(*[p[1] for p in ptrs]),
                                                       _lib.CM_MEM_DEVICE if dev else _lib.CM_MEM_HOST, handles))
        out = []
        for d, h in zip(dims, handles):
            q = cls.__new__(cls)
            q.ctx, q.lib, q.handle, q._keep = ctx, ctx.lib, C.c_void_p(h), None
            q.N, q.K, q.T = d
            out.append(q)
        return out

    @property
    def nnz(self):
        nnz, nv = C.c_int64(), C.c_int32()
        _lib.check(self.lib.cm_problem_nnz(self.handle, C.byref(nnz), C.byref(nv)))
        return nnz.value, nv.value

    def run(self, mu_prior, beta_prior, shape_prior, rate_prior, phi_prior, phi_cov_prior, update_orders=None,
            uniforms=None, verbose=0, timing=0, **fit_options):
        """cm_caviar_run: results stay in HBM.  Returns the device timings (`timing` = 0: fit totals; 1: + per-kernel stamps of the
        chain and mid-iteration kernels; 2: + all phase stamps -- the stamps cost ~8 % of a fit)."""
        N = self.N
        unknown = set(fit_options) - set(_OPTION_NAMES)
        if unknown:
            raise TypeError("caviar() got an unexpected keyword argument '%s'" % sorted(unknown)[0])
        o = make_options(self.lib, verbose=verbose, timing=timing, **fit_options)
        keep = [_lib.f64(mu_prior, (N,)), _lib.f64(beta_prior, (N,)), _lib.f64(phi_prior, (N, 2)), _lib.f64(phi_cov_prior, (N, 2, 2))]
        pr = _lib.CmPriors(_lib.dptr(keep[0]), _lib.dptr(keep[1]), float(shape_prior), float(rate_prior), _lib.dptr(keep[2]),
                           _lib.dptr(keep[3]))
        if update_orders is not None or uniforms is not None:
            S = o.num_mc_samples
            uo = np.ascontiguousarray(update_orders, dtype=np.int32).reshape(o.iters, N)
            uu = _lib.f64(uniforms, (o.iters, N, S, 2))
            keep += [uo, uu]
            o.rng_mode = _lib.CM_RNG_EXPLICIT
            o.update_orders = uo.ctypes.data_as(C.POINTER(C.c_int32))
            o.uniforms = _lib.dptr(uu)
        tm = _lib.CmTimings()
        with self.ctx.lock:
            _lib.check(self.lib.cm_caviar_run(self.ctx.handle, self.handle, C.byref(pr), C.byref(o), C.byref(tm)))
        self._last = (o.iters, bool(o.save_histories))
        return tm

    def fetch(self, want_lam=True):
        """cm_caviar_fetch -> the reference's 17-tuple (caviar.py:100)."""
        N, K = self.N, self.K
        iters, hist = self._last
        out = dict(mu=np.empty(N), beta=np.empty(N), lam=np.empty((N, K)) if want_lam else None, phi=np.empty((N, 2)),
                   phi_cov=np.empty((N, 2, 2)), z=np.empty(K))
        r = _lib.CmResult()
        for k, v in out.items():
            if v is not None:
                setattr(r, k, _lib.dptr(v))
        hists = [None] * 8
        if hist and iters > 0:
            shapes = dict(mu_hist=(iters, N), beta_hist=(iters, N), lam_hist=(iters, N, K), shape_hist=(iters, K),
                          rate_hist=(iters, K), phi_hist=(iters, N, 2), phi_cov_hist=(iters, N, 2, 2), z_hist=(iters, K))
            hists = []
            for k, shp in shapes.items():
                a = np.empty(shp)
                setattr(r, k, _lib.dptr(a))
                hists.append(a)
        with self.ctx.lock:
            _lib.check(self.lib.cm_caviar_fetch(self.ctx.handle, self.handle, C.byref(r)))
        receptive_fields = None                                      # caviar.py:51
        return (out["mu"], out["beta"], out["lam"], np.float64(r.shape), np.float64(r.rate), out["phi"], out["phi_cov"],
                out["z"], receptive_fields, *hists)

    def close(self):
        if self.handle:
            with self.ctx.lock:
                self.lib.cm_problem_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fetch_batch(problems):
    """cm_caviar_fetch_batch: (mu, beta, None, shape, rate, phi, phi_cov, z) of every problem (one context, no lam, no histories) with one
    device-to-host copy."""
    if not problems:
        return []
    ctx = problems[0].ctx
    n = len(problems)
    outs = [dict(mu=np.empty(q.N), beta=np.empty(q.N), phi=np.empty((q.N, 2)), phi_cov=np.empty((q.N, 2, 2)), z=np.empty(q.K)) for q in problems]
    rs = (_lib.CmResult * n)()
    for r, o in zip(rs, outs):
        for k, v in o.items():
            setattr(r, k, _lib.dptr(v))
    handles = (C.c_void_p * n)(*[q.handle for q in problems])
    with ctx.lock:
        _lib.check(ctx.lib.cm_caviar_fetch_batch(ctx.handle, n, handles, rs))
    return [(o["mu"], o["beta"], None, np.float64(r.shape), np.float64(r.rate), o["phi"], o["phi_cov"], o["z"]) for r, o in zip(rs, outs)]


def _priors_struct(N, mu_prior, beta_prior, shape_prior, rate_prior, phi_prior, phi_cov_prior, keep):
    arrs = [_lib.f64(mu_prior, (N,)), _lib.f64(beta_prior, (N,)), _lib.f64(phi_prior, (N, 2)), _lib.f64(phi_cov_prior, (N, 2, 2))]
    keep += arrs
    return _lib.CmPriors(_lib.dptr(arrs[0]), _lib.dptr(arrs[1]), float(shape_prior), float(rate_prior), _lib.dptr(arrs[2]), _lib.dptr(arrs[3]))


def run_batch(problems, priors, update_orders=None, uniforms=None, verbose=0, timing=0, interleave=True, **fit_options):
    """cm_caviar_run_batch: fit several `Problem`s of ONE context in lock-step (the circuit is the outer grid dimension of every
    launch).  `priors`: one dict (shared) or a list of dicts with keys mu, beta, shape, rate, phi, phi_cov.  Results stay in HBM
    (`Problem.fetch`), each bit-identical to the circuit's own `Problem.run`.  `update_orders` [iters, N] / `uniforms`
    [iters, N, S, 2]: explicit RNG streams, shared by all circuits (which must then have the same N).  `interleave` (default): two
    lock-step batches at a time -- a group of 32 circuits or more is split in two halves that run on two streams of the context, the
    dependency-bound phases of one overlapping the wide phases of the other (cm_options.interleave); False: one batch at a time
    (kernel-level profiling).  Results are identical either way.  Returns the device timings."""
    if not problems:
        return []
    ctx = problems[0].ctx
    if any(q.ctx is not ctx for q in problems):
        raise _lib.CaviarHipError("run_batch: all problems must live on the same context")
    unknown = set(fit_options) - set(_OPTION_NAMES)
    if unknown:
        raise TypeError("caviar() got an unexpected keyword argument '%s'" % sorted(unknown)[0])
    o = make_options(ctx.lib, verbose=verbose, timing=timing, interleave=interleave, **fit_options)
    n = len(problems)
    plist = [priors] * n if isinstance(priors, dict) else list(priors)
    keep = []
    # (one struct per distinct (priors dict, N): a batch that shares one priors dict marshals it once -- 256 times was 1.5 ms of Python per step --
    # and the library then stages the shared arrays once as well)
    made = {}
    structs = []
    for q, pr in zip(problems, plist):
        key = (id(pr), q.N)
        if key not in made:
            made[key] = _priors_struct(q.N, pr["mu"], pr["beta"], pr["shape"], pr["rate"], pr["phi"], pr["phi_cov"], keep)
        structs.append(made[key])
    prs = (_lib.CmPriors * n)(*structs)
    if update_orders is not None or uniforms is not None:
        N, S = problems[0].N, o.num_mc_samples
        if any(q.N != N for q in problems):
            raise ValueError("run_batch: explicit RNG streams need circuits of one size")
        uo = np.ascontiguousarray(update_orders, dtype=np.int32).reshape(o.iters, N)
        uu = _lib.f64(uniforms, (o.iters, N, S, 2))
        keep += [uo, uu]
        o.rng_mode = _lib.CM_RNG_EXPLICIT
        o.update_orders = uo.ctypes.data_as(C.POINTER(C.c_int32))
        o.uniforms = _lib.dptr(uu)
    handles = (C.c_void_p * n)(*[q.handle for q in problems])
    tms = (_lib.CmTimings * n)()
    with ctx.lock:
        _lib.check(ctx.lib.cm_caviar_run_batch(ctx.handle, n, handles, prs, C.byref(o), tms))
    for q in problems:
        q._last = (o.iters, bool(o.save_histories))
    return list(tms)


def caviar(y_psc, I, mu_prior, beta_prior, shape_prior, rate_prior, phi_prior, phi_cov_prior,
           iters=50, num_mc_samples=100, seed=0, y_xcorr_thresh=1e-2, minimum_spike_count=3,
           delay_spont_est=1, msrmp=0.3, scale_factor=0.75, penalty=5e0, save_histories=False,
           max_backtrack_iters=20, tol=0.05, spont_orthogonality=0.1, fn_scan=True, *,
           device_id=0, update_orders=None, uniforms=None, verbose=0, return_timings=False, timing=0):
    """Coordinate-ascent variational inference and isotonic regularisation on one MI355X.

    Positional arguments, keyword names, defaults and the 17-element return value are those of the reference
    (circuitmap/optimise/caviar.py:20-23,100).  Keyword-only extras: `device_id`; `update_orders` [iters, N] and
    `uniforms` [iters, N, S, 2] to replace the seeded threefry stream by explicit RNG streams; `return_timings` with
    `timing` = 0 (fit totals), 1 (+ per-kernel stamps) or 2 (+ all phase stamps).
    """
    ctx = _lib.default_context(device_id)
    prob = Problem(ctx, y_psc, I)
    try:
        tm = prob.run(mu_prior, beta_prior, shape_prior, rate_prior, phi_prior, phi_cov_prior, update_orders=update_orders,
                      uniforms=uniforms, verbose=verbose, timing=timing, iters=iters, num_mc_samples=num_mc_samples, seed=seed,
                      y_xcorr_thresh=y_xcorr_thresh, minimum_spike_count=minimum_spike_count, delay_spont_est=delay_spont_est,
                      msrmp=msrmp, scale_factor=scale_factor, penalty=penalty, save_histories=save_histories,
                      max_backtrack_iters=max_backtrack_iters, tol=tol, spont_orthogonality=spont_orthogonality, fn_scan=fn_scan)
        res = prob.fetch()
    finally:
        prob.close()
    return (res, tm) if return_timings else res
