"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libcaviar_port.so, the compiled (C++ / OpenMP) CPU restatement of the reference's
CAVIaR path (oracle/caviar_port.cpp; built by __graft_entry__.build()).  Same signature, defaults and return value as the reference's
``caviar()`` (circuitmap/optimise/caviar.py:20-23,100; histories are not kept).  Used by bench.py's cpu_baseline leg and tests/ only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcaviar_port.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("oracle/libcaviar_port.so is missing: run __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        _lib.caviar_port.restype = C.c_int
        _lib.caviar_port.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, C.c_int, C.c_int, C.c_uint64, C.c_double,
                                     C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                     dp, dp, dp, dp, dp, dp, dp]
    return _lib


def caviar(y_psc, I, mu_prior, beta_prior, shape_prior, rate_prior, phi_prior, phi_cov_prior, iters=50, num_mc_samples=100, seed=0,
           y_xcorr_thresh=1e-2, minimum_spike_count=3, delay_spont_est=1, msrmp=0.3, scale_factor=0.75, penalty=5e0, save_histories=False,
           max_backtrack_iters=20, tol=0.05, spont_orthogonality=0.1, fn_scan=True, n_threads=0):
    lib = load()
    f64 = lambda a, shp: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shp))
    psc = np.ascontiguousarray(np.asarray(y_psc, dtype=np.float64))
    stim = np.ascontiguousarray(np.asarray(I).astype(np.float64))
    N, K = stim.shape
    T = psc.shape[1]
    ins = [psc, stim, f64(mu_prior, (N,)), f64(beta_prior, (N,)), f64(phi_prior, (N, 2)), f64(phi_cov_prior, (N, 2, 2))]
    out = dict(mu=np.empty(N), beta=np.empty(N), lam=np.empty((N, K)), sr=np.empty(2), phi=np.empty((N, 2)), phi_cov=np.empty((N, 2, 2)), z=np.empty(K))
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib.caviar_port(N, K, T, p(ins[0]), p(ins[1]), p(ins[2]), p(ins[3]), float(shape_prior), float(rate_prior), p(ins[4]), p(ins[5]), int(iters),
                         int(num_mc_samples), int(seed), float(y_xcorr_thresh), float(minimum_spike_count), int(delay_spont_est), float(msrmp),
                         float(scale_factor), float(penalty), int(max_backtrack_iters), float(tol), float(spont_orthogonality), int(bool(fn_scan)),
                         int(n_threads), p(out["mu"]), p(out["beta"]), p(out["lam"]), p(out["sr"]), p(out["phi"]), p(out["phi_cov"]), p(out["z"]))
    if rc != 0:
        raise RuntimeError("caviar_port failed with code %d" % rc)
    return (out["mu"], out["beta"], out["lam"], np.float64(out["sr"][0]), np.float64(out["sr"][1]), out["phi"], out["phi_cov"], out["z"], None) + (None,) * 8
