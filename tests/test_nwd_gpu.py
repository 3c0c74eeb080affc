"""NWD demixer forward pass: HIP kernel (through the C ABI / the NeuralDemixer mirror) against golden vectors produced
by the reference's module and against the float32 PyTorch oracle.  Tolerance: float32 -- the kernel accumulates in a
different order (MFMA, BatchNorm folded into the weights), so agreement is to rounding, 2e-5 of the output scale."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "nwd_forward.npz")
TOL = 2e-5


def scale_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return z, {k[4:]: z[k] for k in z.files if k.startswith("sd__")}


@pytest.mark.parametrize("tag", ["t900", "t800"])
def test_matches_reference_golden(gold, tag):
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    z, state = gold
    dm = NeuralDemixer(state=state)
    dem, raw = dm(z[tag + "_traces"], verbose=False, return_raw=True)
    assert scale_err(raw, z[tag + "_raw"]) < TOL
    assert scale_err(dem, z[tag + "_dem"]) < TOL
    dem100 = dm(z[tag + "_traces"], monotone_filter_start=100, monotone_filter_inplace=False, verbose=False)
    assert scale_err(dem100, z[tag + "_dem_start100"]) < TOL
    # the filter itself is exact: running minimum of the rescaled float64 output
    np.testing.assert_array_equal(dem[:, 499:], np.minimum.accumulate((raw.astype(np.float64) * np.max(z[tag + "_traces"], axis=1)[:, None])[:, 499:], axis=1))
    dm.close()


@pytest.mark.parametrize("T", [900, 640, 1000, 777])
def test_matches_oracle_random_weights(T):
    from oracle import nwd_oracle as no
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    state = no.random_state(seed=T)
    rng = np.random.default_rng(T)
    traces = rng.normal(0, 0.05, (37, T)) + np.exp(-np.abs(np.arange(T)[None, :] - rng.uniform(0, T, (37, 1))) / 40.0) * rng.uniform(0.5, 4, (37, 1))
    dm = NeuralDemixer(state=state)
    dem, raw = dm(traces, verbose=False, return_raw=True)
    odem, oraw = no.demix(traces, state, return_raw=True)
    assert np.mean(oraw > 0) > 0.2                           # a non-trivial output (not killed by the last ReLU)
    assert scale_err(raw, oraw) < TOL
    assert scale_err(dem, odem) < TOL
    dm.close()


def test_full_size_properties():
    """K = 2000 traces of 900 samples (the cfg3 batch): row independence, scale equivariance, monotone tail."""
    from oracle import nwd_oracle as no
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    state = no.random_state(seed=5)
    rng = np.random.default_rng(11)
    K, T = 2000, 900
    traces = rng.normal(0, 0.05, (K, T)) + np.exp(-np.abs(np.arange(T)[None, :] - rng.uniform(100, 500, (K, 1))) / 60.0) * rng.uniform(0.5, 4, (K, 1))
    dm = NeuralDemixer(state=state)
    dem = dm(traces, verbose=False)
    assert dem.shape == (K, T) and np.all(np.isfinite(dem))
    assert np.all(np.diff(dem[:, 499:], axis=1) <= 0)
    idx = rng.choice(K, 40, replace=False)
    np.testing.assert_array_equal(dm(traces[idx], verbose=False), dem[idx])            # rows are independent
    odem = no.demix(traces[idx], state)
    assert np.mean(odem != 0) > 0.2
    assert scale_err(dem[idx], odem) < TOL
    scaled = dm(traces[idx] * 8.0, verbose=False)                                      # exact power-of-two rescaling
    np.testing.assert_array_equal(scaled, dem[idx] * 8.0)
    dm.close()


def test_edge_cases():
    from oracle import nwd_oracle as no
    from circuitmap_amd import _lib
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    state = no.random_state(seed=2)
    dm = NeuralDemixer(state=state)
    lib = _lib.load()
    tmin, tmax = lib.cm_nwd_min_trace_len(), lib.cm_nwd_max_trace_len()
    assert tmin == 639 and tmax >= 1000
    rng = np.random.default_rng(0)
    for T in (tmin, tmax):
        tr = rng.normal(0, 1, (3, T))
        dem = dm(tr, verbose=False, monotone_filter_start=min(500, T - 1))
        assert scale_err(dem, no.demix(tr, state, monotone_filter_start=min(500, T - 1))) < TOL
    with pytest.raises(_lib.CaviarHipError):
        dm(rng.normal(0, 1, (2, tmin - 1)), verbose=False)
    with pytest.raises(_lib.CaviarHipError):
        dm(rng.normal(0, 1, (2, tmax + 1)), verbose=False)
    with pytest.raises(_lib.CaviarHipError):
        dm(rng.normal(0, 1, (2, 900)), verbose=False, monotone_filter_start=0)
    assert dm(np.zeros((0, 900)), verbose=False).shape == (0, 900)
    # filter disabled when it starts beyond the trace; single trace; negative-only trace (tmax < 0)
    tr = rng.normal(0, 1, (1, 900))
    assert scale_err(dm(tr, verbose=False, monotone_filter_start=900), no.demix(tr, state, monotone_filter_start=900)) < TOL
    neg = -np.abs(rng.normal(1, 0.2, (2, 900)))
    assert scale_err(dm(neg, verbose=False), no.demix(neg, state)) < TOL
    # a zero trace divides by zero, as in the reference: NaN everywhere, without disturbing the other rows
    z = rng.normal(0, 1, (3, 900)); z[1] = 0.0
    with np.errstate(all="ignore"):
        o = no.demix(z, state)
    d = dm(z, verbose=False)
    assert np.all(np.isnan(d[1])) and np.all(np.isnan(o[1]))
    assert scale_err(d[[0, 2]], o[[0, 2]]) < TOL
    dm.close()


def test_device_resident_path_feeds_caviar():
    """demix on device buffers (cm_mem_* helpers, no torch), then hand the demixed traces to cm_problem_create as device
    memory: same bits as the host-buffer path at both stages."""
    import ctypes as C
    from oracle import nwd_oracle as no
    from circuitmap_amd import _lib
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    state = no.random_state(seed=3)
    rng = np.random.default_rng(3)
    N, K, T = 20, 64, 900
    tr = np.ascontiguousarray(rng.normal(0, 1, (K, T)))
    stim = np.zeros((N, K))
    for k in range(K):
        stim[rng.choice(N, 3, replace=False), k] = rng.choice([45.0, 55.0, 65.0])
    dm = NeuralDemixer(state=state)
    ctx = dm.ctx
    host = dm(tr, verbose=False)
    d_in, d_out, d_stim = ctx.alloc(tr.nbytes), ctx.alloc(tr.nbytes), ctx.alloc(stim.nbytes)
    ctx.upload(d_in, tr)
    ctx.upload(d_stim, stim)
    ms = dm.demix_device(d_in, K, T, d_out)
    assert ms > 0
    np.testing.assert_array_equal(ctx.download(d_out, np.empty_like(tr)), host)
    lib = _lib.load()
    for ptr_psc, ptr_stim, kind in ((d_out, d_stim, _lib.CM_MEM_DEVICE), (host.ctypes.data, stim.ctypes.data, _lib.CM_MEM_HOST)):
        prob = C.c_void_p()
        _lib.check(lib.cm_problem_create(ctx.handle, N, K, T, C.c_void_p(ptr_psc), C.c_void_p(ptr_stim), kind, C.byref(prob)))
        nnz, nv = C.c_int64(), C.c_int32()
        _lib.check(lib.cm_problem_nnz(prob, C.byref(nnz), C.byref(nv)))
        assert nnz.value == 3 * K and nv.value == 4
        lib.cm_problem_destroy(prob)
    for p in (d_in, d_out, d_stim):
        ctx.free(p)
    dm.close()
