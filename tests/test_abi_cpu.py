"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that include/*.h
declare, reports the reference's defaults, and refuses loudly to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from circuitmap_amd import _lib
    return _lib.load()


HEADERS = ("caviar_hip.h", "nwd_hip.h", "sim_hip.h")


def header_text():
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in HEADERS)
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def declared_functions():
    return sorted(set(re.findall(r"\b(cm_[a-z0-9_]+)\s*\(", header_text())))


def test_header_symbols_exported(lib):
    from circuitmap_amd import _lib
    names = declared_functions()
    assert len(names) >= 17
    for name in names:
        assert hasattr(lib, name), "libcaviar_hip.so does not export %s" % name
        assert name in _lib.SYMBOLS, "circuitmap_amd/_lib.py has no ctypes prototype for %s" % name
    assert sorted(_lib.SYMBOLS) == names


def test_struct_layouts_match_header():
    """Field order of the ctypes mirrors == field order of the C structs (a silent mismatch would corrupt arguments)."""
    from circuitmap_amd import _lib
    text = header_text()
    for cname, cls in [("cm_priors", _lib.CmPriors), ("cm_options", _lib.CmOptions), ("cm_result", _lib.CmResult),
                       ("cm_timings", _lib.CmTimings), ("cm_nwd_block", _lib.CmNwdBlock), ("cm_nwd_params", _lib.CmNwdParams),
                       ("cm_sim_options", _lib.CmSimOptions), ("cm_sim_result", _lib.CmSimResult)]:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", re.sub(r"\[[^\]]*\]", "", part))[-1])
        assert fields == [f for f, _ in cls._fields_], cname


def test_defaults_are_the_reference_defaults(lib):
    """cm_options_default == keyword defaults of caviar() (reference circuitmap/optimise/caviar.py:21-23)."""
    from circuitmap_amd import _lib
    o = _lib.CmOptions()
    lib.cm_options_default(C.byref(o))
    ref = dict(iters=50, num_mc_samples=100, seed=0, y_xcorr_thresh=1e-2, minimum_spike_count=3, delay_spont_est=1, msrmp=0.3,
               scale_factor=0.75, penalty=5e0, save_histories=0, max_backtrack_iters=20, tol=0.05, spont_orthogonality=0.1, fn_scan=1)
    for k, v in ref.items():
        assert getattr(o, k) == v, k
    assert o.rng_mode == _lib.CM_RNG_THREEFRY
    assert o.interleave == 0 and o.timing == 0 and o.verbose == 0          # two lock-step batches at a time is the default of the batch entry points


def test_host_side_pava_matches_golden(lib):
    from circuitmap_amd import _lib
    g = np.load(os.path.join(ROOT, "tests", "golden", "pava.npz"))
    for y, o in zip(g["y"], g["out"]):
        y, o = np.ascontiguousarray(y[~np.isnan(y)]), o[~np.isnan(o)]
        out = C.c_double()
        assert lib.cm_pava_last(_lib.dptr(y), len(y), C.byref(out)) == 0
        assert out.value == o[-1]                      # literal pool walk: bit-exact with the reference's pava.py
    assert lib.cm_pava_last(None, 0, None) != 0         # argument errors are codes, not crashes


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from circuitmap_amd import _lib
    assert lib.cm_device_count() == 0
    with pytest.raises(_lib.CaviarHipError, match="no CPU fallback"):
        _lib.Context(0)
    import circuitmap_amd as cm
    sim_psc, stim = np.zeros((4, 5)), np.zeros((2, 4))
    with pytest.raises(_lib.CaviarHipError):
        cm.Model(2).fit(sim_psc, stim, method="caviar")


def test_model_defaults_and_dispatch():
    """Priors of Model == reference circuitmap/model.py:24-31; unknown method raises like model.py:44."""
    import circuitmap_amd as cm
    m = cm.Model(7)
    assert np.array_equal(m.state["mu"], np.zeros(7)) and np.array_equal(m.state["beta"], 10 * np.ones(7))
    assert m.state["shape"] == 1.0 and m.state["rate"] == 0.1
    assert np.array_equal(m.state["phi"], np.c_[0.1 * np.ones(7), 5 * np.ones(7)])
    assert m.state["phi_cov"].shape == (7, 2, 2) and np.array_equal(m.state["phi_cov"][3], np.diag([0.1, 1.0]))
    assert np.array_equal(m.state["alpha"], 0.25 * np.ones(7))
    with pytest.raises(Exception):
        m.fit(None, None, method="bogus")
    with pytest.raises(NotImplementedError):
        m.fit(None, None, method="mbcs")
    with pytest.raises(NotImplementedError, match="UnboundLocalError"):
        m.fit(None, None, method="cavi_sns")


def test_reference_cavi_sns_cannot_run():
    """Why method='cavi_sns' has no MI355X counterpart: in the reference's update_lam the loop body unpacks the carried key as
    `current_key` and then evaluates `jax.random.split(key)` while assigning `key` in the same scope (cavi_sns.py:138-146).
    The same scoping, restated, raises UnboundLocalError in any Python -- there is no behaviour to reproduce."""
    def update_lam(key):
        def body_fun(m, carry):
            lam_vector, current_key = carry
            key, key_next = (key, key)          # noqa: F821,F841  (shape of the reference statement: `key, key_next = split(key)`)
            return lam_vector, key_next
        return body_fun(0, (None, key))
    with pytest.raises(UnboundLocalError):
        update_lam(1)


def test_simulate_contract():
    """Seeded generator: same keys/shapes as the reference's simulate(), blockwise design invariants."""
    import circuitmap_amd as cm
    sim = cm.simulate(N=40, trials=90, H=5, seed=3)
    sim2 = cm.simulate(N=40, trials=90, H=5, seed=3)
    assert set(sim) == {"weights", "phi_0", "phi_1", "mult_noise", "sigma", "stim_matrix", "psc", "gp_noise", "kernel", "spks",
                        "spk_times", "spont_pscs", "I"}
    assert np.array_equal(sim["psc"], sim2["psc"]) and np.array_equal(sim["stim_matrix"], sim2["stim_matrix"])
    stim = sim["stim_matrix"]
    assert stim.shape == (40, 90) and sim["psc"].shape == (90, 900)
    assert np.all((stim > 0).sum(axis=0) == 5)                       # H targets per trial
    assert set(np.unique(stim)) == {0.0, 45.0, 55.0, 65.0}
    assert int((sim["weights"] > 0).sum()) == 2                      # connection_prob 0.05
    assert np.all(sim["spks"][stim == 0] == 0)


def test_dynamic_lds_caps_are_only_ever_raised(lib):
    """hipFuncAttributeMaxDynamicSharedMemorySize replaces the value in force for a (kernel, device); with several contexts on one
    device the library must never lower it: grow / smaller request / grow again, per device, default floor respected."""
    key = 12345
    assert lib.cm_test_lds_request(0, key, 40 * 1024, 48 * 1024) == 0            # below the static default: no call
    assert lib.cm_test_lds_request(0, key, 130 * 1024, 48 * 1024) == 130 * 1024   # context A raises
    assert lib.cm_test_lds_request(0, key, 110 * 1024, 48 * 1024) == 0            # context B (empty cache) asks for less: nothing is lowered
    assert lib.cm_test_lds_request(0, key, 130 * 1024, 48 * 1024) == 0            # A again: still covered
    assert lib.cm_test_lds_request(0, key, 150 * 1024, 48 * 1024) == 150 * 1024   # genuine growth
    assert lib.cm_test_lds_request(1, key, 110 * 1024, 48 * 1024) == 110 * 1024   # another device has its own value
    assert lib.cm_test_lds_request(0, key + 1, 60 * 1024, 0) == 60 * 1024         # another kernel too


def test_dynamic_lds_raise_is_atomic_under_contention(lib):
    """the runtime call that raises a kernel's dynamic-LDS cap happens UNDER the table's lock, so that two host threads
    (contexts on one GPU) cannot apply their values in the opposite order of their table updates.  8 threads x 50 interleaved, growing
    requests against a recording stand-in for hipFuncSetAttribute (20 us per call, to widen any window): the value in force at the end is
    the largest request, and the table agrees with it (a repeat of that request applies nothing)."""
    out = (C.c_uint64 * 4)()
    assert lib.cm_test_lds_race(777, 8, 50, 50 * 1024, 64, 0, out) == 0
    assert out[0] == out[1] == 50 * 1024 + (8 * 50 - 1) * 64
    assert 1 <= out[2] <= 400 and out[3] == 0
    # a value the runtime refuses is not recorded: the table keeps the last accepted one
    assert lib.cm_test_lds_race(778, 4, 20, 10 * 1024, 1024, 40 * 1024, out) == 0
    assert out[0] == 40 * 1024 and out[3] == 0


def test_abi_version_handshake(lib):
    """cm_options carries the ABI version of the header the caller was built against; the library refuses any other (the
    cm_timings layout changed without a version bump)."""
    from circuitmap_amd import _lib
    text = header_text()
    assert int(re.search(r"#define CM_ABI_VERSION (\d+)", text).group(1)) == _lib.CM_ABI_VERSION == lib.cm_abi_version()
    o = _lib.CmOptions()
    lib.cm_options_default(C.byref(o))
    assert o.abi_version == _lib.CM_ABI_VERSION
    assert b"ABI %d" % _lib.CM_ABI_VERSION in lib.cm_version()
