"""NumPy/SciPy-backed stand-in for the small `jax` surface that the reference's
``circuitmap/optimise/caviar.py`` and ``pava.py`` import.  BUILD-CONTAINER TOOLING ONLY: used by
``tests/golden/make_golden.py`` to execute those two reference source files (loaded by path from
/root/reference, never copied) and record golden input/output vectors.  jax itself is not
installable here, so the third-party arithmetic is SUBSTITUTED:

    jax.random.*            -> oracle/threefry.py (Threefry-2x32-20 + jax-0.3 conventions)
    jax.scipy.special.ndtr  -> scipy.special.ndtr      (both Cephes-derived)
    jax.scipy.special.ndtri -> scipy.special.ndtri
    jnp.linalg.inv          -> LAPACK getrf/getri via numpy
    jax.nn.sigmoid          -> scipy.special.expit
    jit / vmap / lax loops  -> plain Python loops (eager)

What the fixtures therefore pin is the reference's own control flow, operation order and quirks;
what they cannot pin is jax's last-ulp numerics.
"""
import sys
import types

import numpy as np
import scipy.special as sps

from oracle import threefry as _tf


class _At:
    def __init__(self, arr):
        self.arr = arr

    def __getitem__(self, idx):
        return _AtIdx(self.arr, idx)


class _AtIdx:
    def __init__(self, arr, idx):
        self.arr, self.idx = arr, idx

    def set(self, val):
        out = np.array(self.arr, copy=True).view(JArray)
        out[self.idx] = val
        return out


class JArray(np.ndarray):
    """ndarray with jax's functional `.at[idx].set(v)` update."""
    __array_priority__ = 100.0

    @property
    def at(self):
        return _At(self)


def _w(x):
    if isinstance(x, tuple):
        return tuple(_w(e) for e in x)
    return np.asarray(x).view(JArray)


def _scalarise(x):
    """Python scalars in loop carries become immutable NumPy scalars (jax turns them into arrays)."""
    if isinstance(x, bool):
        return np.bool_(x)
    if isinstance(x, int):
        return np.int64(x)
    if isinstance(x, float):
        return np.float64(x)
    if isinstance(x, tuple):
        return tuple(_scalarise(e) for e in x)
    return x


def _tree_stack(items):
    first = items[0]
    if isinstance(first, tuple):
        return tuple(_tree_stack([it[i] for it in items]) for i in range(len(first)))
    return _w(np.stack([np.asarray(it) for it in items]))


# ------------------------------------------------------------------ transforms / control flow
def jit(fn=None, static_argnums=None, **kw):
    if fn is None:
        return lambda f: f
    return fn


def vmap(fn, in_axes=0):
    def mapped(*args):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        n = next(np.asarray(a).shape[ax] for a, ax in zip(args, axes) if ax is not None)
        outs = []
        for i in range(n):
            sl = [a if ax is None else _w(np.take(np.asarray(a), i, axis=ax)) for a, ax in zip(args, axes)]
            outs.append(fn(*sl))
        return _tree_stack(outs)
    return mapped


def fori_loop(lo, hi, body, init):
    carry = _scalarise(init)
    for i in range(int(lo), int(hi)):
        carry = body(i, carry)
    return carry


def while_loop(cond, body, init):
    carry = _scalarise(init)
    while bool(cond(carry)):
        carry = body(carry)
    return carry


def scan(f, init, xs):
    carry = init
    ys = []
    for x in np.asarray(xs):
        carry, y = f(carry, x)
        ys.append(y)
    return carry, _tree_stack(ys)


# ------------------------------------------------------------------ jax.numpy
def _unique(x, size=None):
    u = np.unique(np.asarray(x))
    if size is not None:
        if len(u) >= size:
            u = u[:size]
        else:
            u = np.concatenate([u, np.full(size - len(u), u[0] if len(u) else 0)])
    return _w(u)


def _array(x, dtype=None, copy=True):
    return _w(np.array(x, dtype=dtype, copy=True))


def _mk_jnp():
    m = types.ModuleType("jax.numpy")
    passthrough = ["sum", "square", "any", "all", "logical_and", "logical_or", "isnan", "diag", "eye", "arange",
                   "mod", "expand_dims", "mean", "log", "exp", "nan_to_num", "zeros", "zeros_like", "where",
                   "correlate", "ones"]
    for name in passthrough:
        fn = getattr(np, name)
        setattr(m, name, (lambda f: (lambda *a, **k: _w(f(*a, **k))))(fn))
    m.array = _array
    m.asarray = _w
    m.unique = _unique
    m.int32 = np.int32
    m.float64 = np.float64
    m.ndarray = JArray
    la = types.ModuleType("jax.numpy.linalg")
    la.inv = lambda a: _w(np.linalg.inv(np.asarray(a)))
    m.linalg = la
    return m, la


# ------------------------------------------------------------------ jax.random
def _mk_random():
    m = types.ModuleType("jax.random")
    m.PRNGKey = lambda seed: _w(_tf.prng_key(seed))
    m.split = lambda key, num=2: _w(_tf.split(np.asarray(key), num))

    def uniform(key, shape=(), dtype=np.float64, minval=0.0, maxval=1.0):
        return _w(_tf.uniform_f64(np.asarray(key), tuple(shape)))

    def choice(key, a, shape=(), replace=True, p=None, axis=0):
        assert isinstance(a, (int, np.integer)) and not replace and p is None
        return _w(_tf.permutation(np.asarray(key), int(a))[:int(np.prod(shape))])

    m.uniform = uniform
    m.choice = choice
    return m


def install():
    """Register the stand-in modules as `jax`, `jax.numpy`, ... in sys.modules."""
    jax = types.ModuleType("jax")
    jnp, la = _mk_jnp()
    lax = types.ModuleType("jax.lax")
    lax.scan, lax.while_loop, lax.fori_loop = scan, while_loop, fori_loop
    nn = types.ModuleType("jax.nn")
    nn.sigmoid = lambda x: _w(sps.expit(x))
    jsp = types.ModuleType("jax.scipy")
    jsps = types.ModuleType("jax.scipy.special")
    jsps.ndtr = lambda x: _w(sps.ndtr(x))
    jsps.ndtri = lambda x: _w(sps.ndtri(x))
    jsp.special = jsps
    rnd = _mk_random()

    class _Config:
        def update(self, *a, **k):
            pass

    jax.numpy, jax.lax, jax.nn, jax.scipy, jax.random = jnp, lax, nn, jsp, rnd
    jax.jit, jax.vmap, jax.config = jit, vmap, _Config()
    jax.devices = lambda *a: ["cpu0"]
    jax.device_put = lambda x, dev=None: _w(x)
    for name, mod in [("jax", jax), ("jax.numpy", jnp), ("jax.numpy.linalg", la), ("jax.lax", lax),
                      ("jax.nn", nn), ("jax.scipy", jsp), ("jax.scipy.special", jsps), ("jax.random", rnd)]:
        sys.modules[name] = mod
    return jax
