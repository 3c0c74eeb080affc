#!/usr/bin/env python3
"""Kernels of a rocprofv3 rocpd database from the LAST dispatch whose name contains `anchor` on: start offset (us from the anchor), duration, name.
    python tools/rocpd_window.py x_results.db k_init_state 60"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
ncol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = sorted(cur.execute("select %s, start, end from kernels" % ncol).fetchall(), key=lambda r: r[1])
idx = max(i for i, r in enumerate(rows) if anchor in r[0])
t0 = rows[idx][1]
for name, s, e in rows[max(0, idx - 3):idx + n]:
    nm = re.sub(r"\(.*\)$", "", name).replace("void ", "")
    print("%9.1f us  %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, nm[:60]))
