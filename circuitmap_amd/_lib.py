"""ctypes binding of libcaviar_hip.so (include/caviar_hip.h, include/nwd_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or no gfx950 device is
visible, loading / context creation raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAVIAR_HIP_LIB", os.path.join(_HERE, "libcaviar_hip.so"))   # override: A/B builds of the same ABI

CM_OK = 0
CM_MEM_HOST, CM_MEM_DEVICE = 0, 1
CM_RNG_THREEFRY, CM_RNG_EXPLICIT = 0, 1
CM_MAX_VALS = 64
CM_ABI_VERSION = 4          # include/caviar_hip.h; checked against the library at load()

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class CmPriors(C.Structure):
    _fields_ = [("mu", _dp), ("beta", _dp), ("shape", C.c_double), ("rate", C.c_double), ("phi", _dp), ("phi_cov", _dp)]


class CmOptions(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("iters", C.c_int32), ("num_mc_samples", C.c_int32), ("seed", C.c_uint64), ("y_xcorr_thresh", C.c_double),
                ("minimum_spike_count", C.c_double), ("delay_spont_est", C.c_int32), ("msrmp", C.c_double),
                ("scale_factor", C.c_double), ("penalty", C.c_double), ("save_histories", C.c_int32),
                ("max_backtrack_iters", C.c_int32), ("tol", C.c_double), ("spont_orthogonality", C.c_double),
                ("fn_scan", C.c_int32), ("rng_mode", C.c_int32), ("update_orders", _ip), ("uniforms", _dp),
                ("verbose", C.c_int32), ("timing", C.c_int32), ("interleave", C.c_int32)]


class CmStimTriple(C.Structure):
    _fields_ = [("trial", C.c_int32), ("neuron", C.c_int32), ("power", C.c_float)]


class CmInputs(C.Structure):
    _fields_ = [("psc_dtype", C.c_int32), ("stim_layout", C.c_int32), ("psc", C.c_void_p), ("stim", C.c_void_p), ("n_triples", C.c_int64)]


CM_DTYPE_F64, CM_DTYPE_F32 = 0, 1
CM_STIM_DENSE_F64, CM_STIM_DENSE_F32, CM_STIM_TRIPLES = 0, 1, 2


def compact_inputs(psc, stim, keep):
    """cm_inputs for one circuit from host arrays: psc as float32, the stimulus matrix as (trial, neuron, power) triples of its non-zero entries.
    Exact when the values are float32-representable (recorded traces and laser powers are); `keep` receives the buffers the struct points into."""
    p32 = np.ascontiguousarray(psc, dtype=np.float32)
    st = np.asarray(stim)
    nz_n, nz_k = np.nonzero(st)
    tr = np.empty(len(nz_n), dtype=np.dtype([("trial", np.int32), ("neuron", np.int32), ("power", np.float32)]))
    tr["trial"], tr["neuron"], tr["power"] = nz_k, nz_n, st[nz_n, nz_k]
    keep += [p32, tr]
    return CmInputs(CM_DTYPE_F32, CM_STIM_TRIPLES, p32.ctypes.data, tr.ctypes.data if len(tr) else None, len(tr))


class CmResult(C.Structure):
    _fields_ = [("mu", _dp), ("beta", _dp), ("lam", _dp), ("shape", C.c_double), ("rate", C.c_double), ("phi", _dp),
                ("phi_cov", _dp), ("z", _dp), ("mu_hist", _dp), ("beta_hist", _dp), ("lam_hist", _dp), ("shape_hist", _dp),
                ("rate_hist", _dp), ("phi_hist", _dp), ("phi_cov_hist", _dp), ("z_hist", _dp)]


class CmTimings(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("iters_ms", C.c_double), ("fn_scan_ms", C.c_double), ("solve_ms", C.c_double),
                ("sweep_indep_ms", C.c_double), ("sweep_seq_ms", C.c_double), ("mid_ms", C.c_double), ("post_ms", C.c_double), ("mc_dead_ms", C.c_double), ("fact_ms", C.c_double),
                ("iters", C.c_int32), ("batch", C.c_int32), ("last_active", C.c_int32), ("sum_active", C.c_int64), ("sum_seq_rows", C.c_int64), ("sum_seq_records", C.c_int64),
                ("fact_launches", C.c_int64), ("fact_tile_updates", C.c_int64), ("fact_panel_tiles", C.c_int64),
                ("fact_tri_tiles", C.c_int64), ("fact_products", C.c_int64), ("inv_products", C.c_int64), ("tiled_ms", C.c_double), ("tiled_solves", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_fp = C.POINTER(C.c_float)


class CmNwdBlock(C.Structure):
    _fields_ = [("weight", _fp), ("bias", _fp), ("bn_weight", _fp), ("bn_bias", _fp), ("bn_mean", _fp), ("bn_var", _fp)]


class CmNwdParams(C.Structure):
    _fields_ = [("dblock", CmNwdBlock * 4), ("ublock", CmNwdBlock * 4), ("conv", CmNwdBlock), ("bn_eps", C.c_float)]


CM_SIM_MAX_POWERS = 16


class CmSimOptions(C.Structure):
    _fields_ = [("N", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("trials", C.c_int32), ("nreps", C.c_int32),
                ("n_powers", C.c_int32), ("powers", C.c_double * CM_SIM_MAX_POWERS),
                ("connection_prob", C.c_double), ("min_latency", C.c_double), ("gamma_beta", C.c_double), ("sigma", C.c_double),
                ("frac_strongly_connected", C.c_double),
                ("strong_weight_lower", C.c_double), ("strong_weight_upper", C.c_double), ("weak_exp_mean", C.c_double), ("min_weight", C.c_double),
                ("phi_0_lower", C.c_double), ("phi_0_upper", C.c_double), ("phi_1_lower", C.c_double), ("phi_1_upper", C.c_double),
                ("mult_noise_log_var", C.c_double),
                ("tau_r_min", C.c_double), ("tau_r_max", C.c_double), ("tau_delta_min", C.c_double), ("tau_delta_max", C.c_double),
                ("gp_scale", C.c_double), ("gp_lengthscale", C.c_double), ("spont_prob", C.c_double), ("max_power_min_spike_rate", C.c_double),
                ("seed", C.c_uint64)]


class CmSimResult(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("psc", "stim_matrix", "weights", "phi_0", "phi_1", "spks", "spk_times", "I", "gp_noise",
                                          "spont_pscs", "tau_r", "tau_d")]


# every symbol include/*.h declare: (restype, argtypes)
SYMBOLS = {
    "cm_last_error": (C.c_char_p, []),
    "cm_version": (C.c_char_p, []),
    "cm_abi_version": (C.c_int, []),
    "cm_device_count": (C.c_int, []),
    "cm_options_default": (None, [C.POINTER(CmOptions)]),
    "cm_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cm_ctx_destroy": (C.c_int, [C.c_void_p]),
    "cm_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "cm_ctx_release_cache": (C.c_int, [C.c_void_p]),
    "cm_mem_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "cm_mem_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cm_mem_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "cm_problem_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "cm_problem_create_batch": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]),
    "cm_problem_create_batch_ex": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(CmInputs),
                                             C.c_int, C.POINTER(C.c_void_p)]),
    "cm_caviar_fit_batch_ex": (C.c_int, [C.c_int32, _ip, _ip, _ip, C.POINTER(CmInputs), C.POINTER(CmPriors), C.POINTER(CmOptions),
                                         _ip, C.c_int32, C.POINTER(CmResult), C.POINTER(CmTimings)]),
    "cm_caviar_fit_batch_release": (C.c_int, []),
    "cm_caviar_fetch_batch": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(CmResult)]),
    "cm_problem_destroy": (C.c_int, [C.c_void_p]),
    "cm_problem_nnz": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "cm_caviar_run": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CmPriors), C.POINTER(CmOptions), C.POINTER(CmTimings)]),
    "cm_caviar_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CmResult)]),
    "cm_caviar_fit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, C.POINTER(CmPriors), C.POINTER(CmOptions),
                                C.POINTER(CmResult), C.POINTER(CmTimings)]),
    "cm_caviar_run_batch": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(CmPriors), C.POINTER(CmOptions), C.POINTER(CmTimings)]),
    "cm_caviar_fit_batch": (C.c_int, [C.c_int32, _ip, _ip, _ip, C.POINTER(_dp), C.POINTER(_dp), C.POINTER(CmPriors), C.POINTER(CmOptions),
                                      _ip, C.c_int32, C.POINTER(CmResult), C.POINTER(CmTimings)]),
    "cm_k_sigma": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp]),
    "cm_k_spont": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp, C.POINTER(C.c_uint8), _dp, C.c_int32, C.c_int32, C.c_double,
                             C.c_double, C.c_double, C.c_double, _dp, _dp]),
    "cm_k_lam_sweep": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp, C.POINTER(C.c_uint8), _dp, _dp, C.c_double, C.c_double, _dp, _dp,
                                 C.POINTER(C.c_uint32), C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32]),
    "cm_k_prologue": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp]),
    "cm_pava_last": (C.c_int, [_dp, C.c_int32, _dp]),
    "cm_k_block_update_mu": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp]),
    "cm_k_update_phi": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp]),
    "cm_k_reconnect": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double]),
    "cm_k_rng_streams": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, _ip, _dp]),
    "cm_k_ndtri": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp]),
    "cm_k_mfma_peak": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "cm_k_event_pair_overhead": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
    "cm_k_solve_bench": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cm_test_lds_request": (C.c_int64, [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64]),
    "cm_test_lds_race": (C.c_int, [C.c_uint64, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    # include/nwd_hip.h
    "cm_nwd_create": (C.c_int, [C.c_void_p, C.POINTER(CmNwdParams), C.POINTER(C.c_void_p)]),
    "cm_nwd_destroy": (None, [C.c_void_p]),
    "cm_nwd_max_trace_len": (C.c_int, []),
    "cm_nwd_min_trace_len": (C.c_int, []),
    # include/sim_hip.h
    "cm_sim_options_default": (None, [C.POINTER(CmSimOptions)]),
    "cm_simulate": (C.c_int, [C.c_void_p, C.POINTER(CmSimOptions), C.c_int, C.POINTER(CmSimResult), C.c_int, C.POINTER(C.c_float)]),
    "cm_nwd_demix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_int, C.POINTER(C.c_float)]),
}

_lib = None


class CaviarHipError(RuntimeError):
    pass


def load():
    """Load libcaviar_hip.so (built in-tree by __graft_entry__.build()); raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CaviarHipError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(there is no CPU fallback)" % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own copy of libamdhip64, and whichever library is loaded
        # first decides which copy the process uses.  Device pointers and streams are exchanged with torch (bench.py, batch.py,
        # the demixer), so torch goes first whenever it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.cm_abi_version() != CM_ABI_VERSION:
            raise CaviarHipError("%s is ABI %d, this binding expects %d: rebuild the library" % (LIB_PATH, lib.cm_abi_version(), CM_ABI_VERSION))
        _lib = lib
    return _lib


def check(rc):
    if rc != CM_OK:
        raise CaviarHipError("libcaviar_hip error %d: %s" % (rc, load().cm_last_error().decode()))


def dptr(a):
    return a.ctypes.data_as(_dp)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Context:
    """One GPU + one HIP stream.  `stream` may be a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device_id=0, stream=None):
        self.lib = load()
        self.handle = C.c_void_p()
        self.lock = threading.RLock()       # calls on one context must be serialised (caviar_hip.h): Problem.run / fetch take it
        check(self.lib.cm_ctx_create(int(device_id), C.c_void_p(stream) if stream else None, C.byref(self.handle)))

    def synchronize(self):
        with self.lock:
            check(self.lib.cm_ctx_synchronize(self.handle))

    def release_cache(self):
        """Return the device memory the context keeps from destroyed problems (its pool) to the driver."""
        with self.lock:
            check(self.lib.cm_ctx_release_cache(self.handle))

    # device buffers without torch (cm_mem_*): addresses are plain ints.  Every entry into the library on this context takes the
    # context's lock (caviar_hip.h: calls on one cm_ctx must be serialised; contexts are shared with worker threads in fit_batch / bench).
    def alloc(self, nbytes):
        p = C.c_void_p()
        with self.lock:
            check(self.lib.cm_mem_alloc(self.handle, int(nbytes), C.byref(p)))
        return p.value or 0

    def free(self, ptr):
        with self.lock:
            check(self.lib.cm_mem_free(self.handle, C.c_void_p(ptr)))

    def upload(self, ptr, array):
        a = np.ascontiguousarray(array)
        with self.lock:
            check(self.lib.cm_mem_copy(self.handle, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1))

    def download(self, ptr, array):
        assert array.flags["C_CONTIGUOUS"]
        with self.lock:
            check(self.lib.cm_mem_copy(self.handle, array.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), array.nbytes, 2))
        return array

    def close(self):
        with self.lock:
            if self.handle:
                self.lib.cm_ctx_destroy(self.handle)
                self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_default_ctx_lock = threading.Lock()


def default_context(device_id=0):
    with _default_ctx_lock:
        if device_id not in _default_ctx:
            _default_ctx[device_id] = Context(device_id)
        return _default_ctx[device_id]
