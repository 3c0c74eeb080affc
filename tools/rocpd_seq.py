#!/usr/bin/env python3
"""Dispatch-by-dispatch listing (start offset, duration, gap to the previous dispatch) of the LAST `n` kernels of a rocprofv3 rocpd database:
    python tools/rocpd_seq.py x_results.db [n] [name filter]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
flt = sys.argv[3] if len(sys.argv) > 3 else ""
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
ncol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = sorted(cur.execute("select %s, start, end from kernels" % ncol).fetchall(), key=lambda r: r[1])
rows = [r for r in rows if flt in r[0]][-n:]
prev = None
for name, s, e in rows:
    nm = re.sub(r"\(.*\)$", "", name).replace("void ", "")
    print("%-34s dur %8.1f us   gap %7.1f us" % (nm[:34], (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
