#!/usr/bin/env python3
"""per-role durations of k_presweep launched with CAVIAR_HIP_SPLIT_PRESWEEP=1 (three launches per iteration: rows, columns, scan)"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
ncol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, start, end from kernels order by start" % ncol).fetchall()
d = [e - s for n, s, e in rows if "k_presweep<true" in n]
for r, nm in enumerate(("rows", "columns", "scan")):
    x = d[r::3]
    late = x[len(x) // 5:]
    print("%-8s launches %d avg %.1f us (steady state %.1f us)" % (nm, len(x), sum(x) / len(x) / 1e3, sum(sorted(late)[: len(late) * 3 // 4]) / (len(late) * 3 // 4) / 1e3))
