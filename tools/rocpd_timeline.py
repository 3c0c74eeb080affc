#!/usr/bin/env python3
"""durations of the launches of one kernel in dispatch order: python tools/rocpd_timeline.py x.db k_iter_mid [first] [count]"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 60
rows = db.execute("select name, start, end from kernels order by start").fetchall()
sel = [(n, s, e) for n, s, e in rows if pat in n]
for n, s, e in sel[first:first + count]:
    print("%-40s %9.1f us" % (n[:40], (e - s) / 1e3))
