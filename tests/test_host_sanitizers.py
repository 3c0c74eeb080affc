"""SURVEY section 5 (sanitizers on the CPU build only): the library's host-only code -- the jax key chain, the literal PAVA walk and
reconnect_spont_cells (circuitmap_amd/csrc/cm_host.cpp, plain C++) -- compiled with g++ -fsanitize=address,undefined into a small
driver (tests/host_asan/main.cpp) and checked against the oracle / the golden vectors; any out-of-bounds access, use after free,
signed overflow or misaligned access aborts the run."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN
from oracle import caviar_oracle as co
from oracle import threefry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asan") / "host_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "host_asan", "main.cpp"), os.path.join(ROOT, "circuitmap_amd", "csrc", "cm_host.cpp"), "-o", out]
    subprocess.run(cmd, check=True)
    return out


def run(exe, text):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([exe], input=text.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    return p.stdout.decode().split()


hx = lambda a: " ".join(float(v).hex() for v in np.ravel(a))
ix = lambda a: " ".join(str(int(v)) for v in np.ravel(a))


def test_pava_walk(exe):
    g = np.load(GOLDEN + "/pava.npz")
    text, want = "", []
    for y, o in zip(g["y"], g["out"]):
        y = y[~np.isnan(y)]
        text += "pava %d %s\n" % (len(y), hx(y))
        want.append(o[~np.isnan(o)][-1])
    got = [float.fromhex(t) for t in run(exe, text)]
    assert got == want                                   # bit-exact with the reference's pava.py


@pytest.mark.parametrize("seed,iters,N", [(0, 3, 50), (12345678901, 2, 7), (3, 5, 2), (1, 1, 2000)])
def test_key_schedule(exe, seed, iters, N):
    toks = run(exe, "keys %d %d %d\n" % (seed, iters, N))
    rounds, keys = int(toks[0]), np.array(toks[1:], dtype=np.uint64).astype(np.uint32).reshape(iters, N + 3, 2)
    assert rounds == int(np.ceil(3 * np.log(N) / np.log(2 ** 32 - 1)))
    key = threefry.prng_key(seed)
    for it in range(iters):
        k = key.copy()
        for r in range(rounds):                           # permutation rounds: key, subkey = split(key)
            k, sub = threefry.split(k)
            assert np.array_equal(keys[it, r], sub)
        k = key.copy()
        for m in range(N):                                # sweep steps: key, key_next = split(key); uniform(key)
            ks = threefry.split(k)
            assert np.array_equal(keys[it, 3 + m], ks[0])
            k = ks[1]
        key = threefry.split(k)[1]                        # update_phi hands back split(key)[1]


def test_reconnect_golden_and_random(exe):
    g = np.load(GOLDEN + "/units.npz")
    cases = [(g["stim"], g["lam"], g["rec_mu_in"], g["bum_beta"], g["rec_z_in"], 0.3, 3.0, (g["rec_mu"], g["rec_beta"], g["rec_lam"], g["rec_z"]))]
    rng = np.random.default_rng(4)
    for trial in range(10):                               # random scenarios against the oracle, incl. single-event reconnections (NaN sem)
        N, K = int(rng.integers(3, 9)), int(rng.integers(12, 40))
        if trial >= 6:                                    # long rows: cells reconnect on 8+ (up to > 128) events -> NumPy's pairwise mean / sem (cm_np_sum)
            N, K = int(rng.integers(3, 6)), int(rng.integers(150, 900))
        powers = np.array([30.0, 50.0, 70.0])[: int(rng.integers(1, 4))]
        stim = np.zeros((N, K))
        for k in range(K):
            stim[rng.choice(N, min(N, 2), replace=False), k] = powers[rng.integers(0, len(powers))]
        lam = rng.random((N, K)) * (stim > 0)
        mu = rng.normal(5, 2, N) * (rng.random(N) > 0.5)
        z = rng.uniform(2, 6, K) * (rng.random(K) > 0.4)
        mc = float(rng.integers(1, 4))
        cases.append((stim, lam, mu, rng.random(N), z, 0.2, mc, None))
    for stim, lam, mu, beta, z, minimax, mc, want in cases:
        N, K = stim.shape
        vals = np.unique(stim)
        if vals[0] != 0:
            vals = np.concatenate([[0.0], vals])
        rows = [np.flatnonzero(stim[n] > 0) for n in range(N)]
        row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
        col = np.concatenate(rows) if rows else np.zeros(0, int)
        grp = np.concatenate([np.searchsorted(vals, stim[n, r]) for n, r in enumerate(rows)])
        cols = [np.flatnonzero(stim[:, k] > 0) for k in range(K)]
        col_ptr = np.concatenate([[0], np.cumsum([len(c) for c in cols])])
        csc_row = np.concatenate(cols)
        lam_c = np.concatenate([lam[n, r] for n, r in enumerate(rows)])
        text = "reconnect %d %d %d %d %s %s %s %s %s %s %s %s %s %s %s\n" % (
            N, K, len(vals), len(col), ix(row_ptr), ix(col), ix(grp), ix(col_ptr), ix(csc_row), hx(mu), hx(beta), hx(lam_c), hx(z),
            float(minimax).hex(), float(mc).hex())
        got = np.array([float.fromhex(t) for t in run(exe, text)])
        gm, gb, gl, gz = got[:N], got[N:2 * N], got[2 * N:2 * N + len(col)], got[2 * N + len(col):]
        if want is None:
            want = co.reconnect_spont_cells(np.zeros(K), stim, lam, mu, beta, z, minimax, mc)
        wl = np.concatenate([np.asarray(want[2])[n, r] for n, r in enumerate(rows)])
        assert np.array_equal(gm, want[0]) and np.array_equal(gl, wl) and np.array_equal(gz, want[3])
        assert np.allclose(gb, want[1], rtol=1e-14, atol=0, equal_nan=True)
