#!/usr/bin/env python3
"""Occupancy probe: what the runtime reports for a kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor) -- diagnostic only.
usage: python tools/occ_probe.py  (on the GPU box; prints registers / LDS of the launch as rocprofv3 recorded them when given a db)"""
import sqlite3
import sys

if len(sys.argv) > 1:
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t]
    print(kd)
    for t in kd[:1]:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
        print(cols)
    t = "kernels" if "kernels" in tabs else kd[0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
    print(t, cols)
    want = [c for c in cols if any(k in c.lower() for k in ("name", "lds", "vgpr", "sgpr", "scratch", "workgroup", "grid", "private", "group_segment"))]
    seen = set()
    for row in db.execute("select %s from %s" % (",".join(want), t)):
        key = tuple(row[:1])
        if ("chol" in str(row[0]) or "inv_" in str(row[0])) and row not in seen:
            seen.add(row)
            if len(seen) < 40:
                print(dict(zip(want, row)))
