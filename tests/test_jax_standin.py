"""The golden vectors come from the reference's own caviar.py / pava.py executed under tests/golden/jax_standin.py (jax is not
installable in this image).  This file pins, symbol by symbol, the stand-in's semantics that those two files rely on to what the
jax documentation states for the same call -- it is as far as the parity pin can be tightened without jax itself
(jax's own last-ulp numerics and PRNG internals stay "parity unpinned")."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import jax_standin  # noqa: E402
from oracle import threefry  # noqa: E402


@pytest.fixture(scope="module")
def jax():
    saved = {k: v for k, v in sys.modules.items() if k == "jax" or k.startswith("jax.")}
    j = jax_standin.install()
    yield j
    for k in [k for k in sys.modules if k == "jax" or k.startswith("jax.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_unique_with_size(jax):
    jnp = jax.numpy
    # jnp.unique(x, size=k): sorted unique values, truncated to k, or padded with fill_value = the MINIMUM unique value (jax docs)
    assert np.array_equal(jnp.unique(np.array([3, 1, 3, 2]), size=3), [1, 2, 3])
    assert np.array_equal(jnp.unique(np.array([3, 1, 3, 2]), size=2), [1, 2])
    assert np.array_equal(jnp.unique(np.array([5, 7, 5]), size=4), [5, 7, 5, 5])
    # its one use, caviar.py:204: "all ids except n" (N - 1 distinct values -> no padding ever happens)
    N = 7
    ids = jnp.arange(N)
    for n in (0, 3, N - 1):
        mask = jnp.unique(jnp.where(ids != n, ids, jnp.mod(n - 1, N)), size=N - 1)
        assert np.array_equal(mask, [i for i in range(N) if i != n])


def test_negative_index_gather_and_masking(jax):
    jnp = jax.numpy
    # caviar.py:181-183: locs = where(stimv == power, Krange, -1); lamv[locs] * (locs >= 0).  Index -1 wraps to the LAST element
    # in jax as in NumPy; the mask then removes it, whatever its value
    lamv = jnp.array([0.1, 0.2, 0.4, 100.0])
    stimv = np.array([5.0, 0.0, 5.0, 0.0])
    locs = jnp.where(stimv == 5.0, jnp.arange(4), -1)
    assert np.array_equal(locs, [0, -1, 2, -1]) and np.array_equal(lamv[locs], [0.1, 100.0, 0.4, 100.0])
    m = locs >= 0
    assert jnp.sum(lamv[locs] * m) / (jnp.sum(m) + 1e-4 * (jnp.sum(m) == 0.)) == pytest.approx(0.25, abs=1e-16)
    none = jnp.where(stimv == 9.0, jnp.arange(4), -1) >= 0
    assert jnp.sum(lamv * 0) / (jnp.sum(none) + 1e-4 * (jnp.sum(none) == 0.)) == 0.0     # empty power group: 0 / 1e-4


def test_nan_to_num_correlate_zeros_like(jax):
    jnp = jax.numpy
    fi = np.finfo(np.float64)
    got = jnp.nan_to_num(np.array([np.nan, np.inf, -np.inf, 1.5]))          # jax defaults: nan -> 0, +-inf -> finfo max / min
    assert np.array_equal(got, [0.0, fi.max, fi.min, 1.5])
    a = np.array([1.0, -2.0, 3.0])
    c = jnp.correlate(a, a)                                                  # default mode 'valid': one element, sum a^2 (caviar.py:30)
    assert c.shape == (1,) and c[0] == 14.0
    assert np.array_equal(jnp.zeros_like(np.array([3, 0])) + 0, [0, 0]) and jnp.zeros_like(np.array([3, 0])).dtype.kind == "i"   # quirk 9


def test_functional_updates_and_loop_carries(jax):
    jnp, lax = jax.numpy, jax.lax
    x = jnp.zeros(3)
    y = x.at[1].set(5.0)
    assert np.array_equal(x, [0, 0, 0]) and np.array_equal(y, [0, 5, 0])     # .at[].set returns a copy
    z = y.at[jnp.array([0, 2])].set(jnp.array([1.0, 2.0]))
    assert np.array_equal(z, [1, 5, 2])
    # Python scalars in a loop carry become (immutable) typed scalars, as jax turns them into 0-d arrays: int stays integer,
    # float is float64 under jax_enable_x64 (caviar.py:12), and the trip count / cond are evaluated eagerly in order
    out = lax.while_loop(lambda c: c[0] < 4, lambda c: (c[0] + 1, c[1] * 0.5, c[2]), (0, 1.0, True))
    assert out[0] == 4 and isinstance(out[0], np.integer) and out[1] == 0.0625 and isinstance(out[1], np.float64)
    acc = lax.fori_loop(2, 5, lambda i, c: c + i, 0)
    assert acc == 2 + 3 + 4
    carry, ys = lax.scan(lambda c, x_: (c + x_, c * x_), 1.0, jnp.arange(4))
    assert carry == 7.0 and np.array_equal(ys, [0.0, 1.0, 4.0, 12.0])
    f = jax.vmap(lambda a, b, s_: a * s_ + b, in_axes=(0, 0, None))
    assert np.array_equal(f(jnp.array([1.0, 2.0]), jnp.array([10.0, 20.0]), 3.0), [13.0, 26.0])
    g = jax.vmap(lambda a: (a + 1, a * 2))(jnp.array([1.0, 2.0]))             # tuple outputs are stacked leaf by leaf
    assert np.array_equal(g[0], [2, 3]) and np.array_equal(g[1], [2, 4])


def test_special_functions_and_inverse(jax):
    sp, nn, la = jax.scipy.special, jax.nn, jax.numpy.linalg
    x = np.array([-8.0, -1.0, 0.0, 0.5, 3.0])
    assert np.allclose(sp.ndtri(sp.ndtr(x)), x, rtol=0, atol=1e-9)
    assert sp.ndtr(0.0) == 0.5 and sp.ndtr(-np.inf) == 0.0 and sp.ndtri(0.5) == 0.0
    assert nn.sigmoid(0.0) == 0.5 and nn.sigmoid(800.0) == 1.0 and nn.sigmoid(-800.0) == 0.0     # saturation without overflow warnings
    A = np.array([[2.0, 1.0], [1.0, 3.0]])
    assert np.allclose(la.inv(A) @ A, np.eye(2), atol=1e-15)


def test_random_layer_is_the_threefry_restatement(jax):
    rnd = jax.random
    key = rnd.PRNGKey(0)
    assert np.array_equal(rnd.split(key), [[4146024105, 967050713], [2718843009, 1272950319]])   # value printed in the jax docs
    # choice(key, N, [N], replace=False) is permutation(key, N) (jax: "without replacement = permutation[:n]"), caviar.py:196
    assert np.array_equal(rnd.choice(key, 11, [11], replace=False), threefry.permutation(np.asarray(key), 11))
    u = rnd.uniform(key, [5, 2])
    assert u.shape == (5, 2) and u.dtype == np.float64 and np.all((u >= 0) & (u < 1))
    assert sorted(rnd.choice(key, 9, [9], replace=False)) == list(range(9))
