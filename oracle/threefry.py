"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the counter-based PRNG the reference consumes.

The reference draws all CAVIaR randomness from ``jax.random`` (PRNGKey / split / uniform / choice;
call sites ``circuitmap/optimise/caviar.py:76,196,209-210,304``).  jax / jaxlib are a third-party
dependency that is absent from ``/root/reference`` and from this image (the reference README says
"JAX versions up to 0.3.15", ``README.md:22``; it is not pinned in ``setup.py:11-19``).  This file
restates the published algorithm (Threefry-2x32, 20 rounds, Salmon et al. SC'11, as used by
``jax._src.prng`` in the 0.3 series) and the jax-0.3 conventions layered on top of it:

* ``PRNGKey(seed) = [seed >> 32, seed & 0xffffffff]``
* ``split(key, n)``: counters ``iota(2n)``, first half -> x0 lane, second half -> x1 lane, outputs
  concatenated and reshaped ``(n, 2)``
* ``random_bits(key, 32, shape)``: counters ``iota(size)`` (padded with one 0 if odd), halves as above
* ``random_bits(key, 64, shape)``: ``2*size`` 32-bit words; hi = first half, lo = second half
* ``uniform`` f64: ``bitcast((bits >> 12) | 0x3FF0000000000000) - 1``
* ``choice(key, N, [N], replace=False)`` = ``permutation`` = ceil(3 ln N / ln(2^32-1)) rounds of
  (split; 32-bit sort keys from the subkey; STABLE sort-by-key).

Pinned by the known-answer vectors in ``tests/test_threefry.py`` (Random123 KATs for
Threefry-2x32-20 and the two values the jax documentation prints for ``PRNGKey(0)``).  The
higher-level conventions cannot be confirmed against a live jax here; every parity test therefore
ALSO feeds explicit RNG streams to both the oracle and the HIP path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""
import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32-20 block function on uint32 arrays (vectorised over counters)."""
    with np.errstate(over="ignore"):
        k0 = _U32(k0)
        k1 = _U32(k1)
        x0 = np.array(x0, dtype=_U32, copy=True)
        x1 = np.array(x1, dtype=_U32, copy=True)
        ks = (k0, k1, _U32(k0 ^ k1 ^ _U32(0x1BD11BDA)))
        x0 += ks[0]
        x1 += ks[1]
        for g in range(5):
            for r in _ROT[g % 2]:
                x0 += x1
                x1 = _rotl(x1, r)
                x1 ^= x0
            x0 += ks[(g + 1) % 3]
            x1 += ks[(g + 2) % 3] + _U32(g + 1)
    return x0, x1


def prng_key(seed):
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=_U32)


def _bits32(key, n):
    """n 32-bit words from counters iota(n), jax `threefry_2x32(key, iota(n))` layout."""
    odd = n % 2
    cnt = np.arange(n + odd, dtype=_U32)
    if odd:
        cnt[-1] = 0
    half = (n + odd) // 2
    y0, y1 = threefry2x32(key[0], key[1], cnt[:half], cnt[half:])
    return np.concatenate([y0, y1])[:n]


def split(key, num=2):
    return _bits32(key, 2 * num).reshape(num, 2)


def random_bits64(key, size):
    w = _bits32(key, 2 * size)
    return (w[:size].astype(np.uint64) << np.uint64(32)) | w[size:].astype(np.uint64)


def uniform_f64(key, shape):
    size = int(np.prod(shape))
    bits = random_bits64(key, size)
    fb = (bits >> np.uint64(12)) | np.uint64(0x3FF0000000000000)
    return (fb.view(np.float64) - 1.0).reshape(shape)


def uniform_f32(key, shape=()):
    size = int(np.prod(shape)) if shape else 1
    bits = _bits32(key, size)
    fb = (bits >> _U32(9)) | _U32(0x3F800000)
    return (fb.view(np.float32) - np.float32(1.0)).reshape(shape)


def permutation(key, n):
    """jax.random.choice(key, n, [n], replace=False) for an integer population."""
    x = np.arange(n)
    rounds = int(np.ceil(3 * np.log(max(1, n)) / np.log(np.iinfo(np.uint32).max)))
    for _ in range(rounds):
        key, sub = split(key)
        sk = _bits32(sub, n)
        x = x[np.argsort(sk, kind="stable")]
    return x


class JaxLikeStream:
    """RNG consumption order of ``caviar()`` (caviar.py:76,196,209-210,227,251,304).

    Per iteration: ``update_order(N)`` uses the carried key itself; each of the N sweep steps does
    ``key, key_next = split(key)``, draws ``uniform(key, [S, 2])`` and carries ``key_next``; then
    ``update_phi`` replaces the key by ``split(key)[1]``.
    """

    def __init__(self, seed):
        self.key = prng_key(seed)

    def update_order(self, it, N):
        return permutation(self.key, N)

    def uniforms(self, it, m, S):
        ks = split(self.key)
        self.key = ks[1]
        return uniform_f64(ks[0], (S, 2))

    def end_iteration(self, it):
        self.key = split(self.key)[1]


class ExplicitStream:
    """Host-supplied streams: update_orders[iters, N] and uniforms[iters, N, S, 2] (indexed by sweep step)."""

    def __init__(self, update_orders, uniforms):
        self.orders = np.asarray(update_orders)
        self.u = np.asarray(uniforms)

    def update_order(self, it, N):
        return self.orders[it]

    def uniforms(self, it, m, S):
        return self.u[it, m]

    def end_iteration(self, it):
        pass


def make_explicit_streams(seed, iters, N, S):
    """Materialise the JaxLikeStream for `iters` iterations as arrays (what the C-ABI accepts as host streams)."""
    st = JaxLikeStream(seed)
    orders = np.zeros((iters, N), dtype=np.int32)
    u = np.zeros((iters, N, S, 2))
    for it in range(iters):
        orders[it] = st.update_order(it, N)
        for m in range(N):
            u[it, m] = st.uniforms(it, m, S)
        st.end_iteration(it)
    return orders, u
