#!/usr/bin/env python3
"""Tiled-solve microbenchmark (C ABI cm_k_solve_bench): B synthetic SPD matrices of n rows through the Cholesky and inverse launches.
    python tools/solve_bench.py [n] [B] [reps]     e.g.  python tools/solve_bench.py 1000 128 5"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from circuitmap_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = _lib.Context(0)
a, b = C.c_float(), C.c_float()
_lib.check(ctx.lib.cm_k_solve_bench(ctx.handle, n, B, reps, C.byref(a), C.byref(b)))
gf = B * (n ** 3 / 3.0) / 1e9
print("n=%d B=%d family=%s: cholesky %.3f ms (%.1f TFLOP/s)  inverse %.3f ms (%.1f TFLOP/s)" % (
    n, B, os.environ.get("CAVIAR_HIP_CHOL_FAMILY", "1"), a.value, gf / a.value, b.value, gf / b.value))
