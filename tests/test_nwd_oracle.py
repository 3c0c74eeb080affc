"""NWD oracle (oracle/nwd_oracle.py) against golden vectors produced by the reference's own module (CPU, no GPU)."""
import os

import numpy as np
import pytest

from oracle import nwd_oracle as no

GOLD = os.path.join(os.path.dirname(__file__), "golden", "nwd_forward.npz")


def load_gold():
    z = np.load(GOLD)
    state = {k[4:]: z[k] for k in z.files if k.startswith("sd__")}
    return z, state


@pytest.mark.parametrize("tag", ["t900", "t800"])
def test_oracle_matches_reference_forward(tag):
    import torch
    z, state = load_gold()
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)       # the fixture was generated single-threaded; PyTorch's conv splits reductions by thread count
    try:
        _check(z, state, tag)
    finally:
        torch.set_num_threads(nthreads)
    # multi-threaded: same result to float32 rounding
    raw = no.demix(z[tag + "_traces"], state, return_raw=True)[1]
    assert np.max(np.abs(raw - z[tag + "_raw"])) <= 1e-5 * np.max(np.abs(z[tag + "_raw"]))


def _check(z, state, tag):
    dem, raw = no.demix(z[tag + "_traces"], state, return_raw=True)
    # same float32 PyTorch kernels underneath: the restatement reproduces the module bit for bit
    np.testing.assert_array_equal(raw.astype(np.float32), z[tag + "_raw"])
    np.testing.assert_array_equal(dem, z[tag + "_dem"])
    np.testing.assert_array_equal(no.demix(z[tag + "_traces"], state, monotone_filter_start=100), z[tag + "_dem_start100"])


def test_monotone_filter_is_running_minimum():
    rng = np.random.default_rng(0)
    a = rng.normal(size=(5, 300))
    f = no.monotone_decay_filter(a, 120)
    np.testing.assert_array_equal(f[:, :120], a[:, :120])
    assert np.all(np.diff(f[:, 119:], axis=1) <= 0)
    np.testing.assert_array_equal(f[:, 119:], np.minimum.accumulate(a[:, 119:], axis=1))


def test_layer_lengths_T900():
    """the lengths the HIP kernel's LDS plan is built on: 449/387, 193/162, 80/65, 32/17, 32, 80, 193, 804"""
    import torch
    state = no.random_state(1)
    x = torch.zeros(1, 1, 900)
    assert no.unet_forward(x, state).shape == (1, 1, 900)
