#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max duration, share) of a rocprofv3 rocpd SQLite database, as CSV on stdout:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.csv
(rocprofv3 on this image writes `--kernel-trace --stats` output as a .db; `--output-format csv` gives the same table directly.)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)            # drop the argument list
    name = re.sub(r"^void ", "", name)
    return name


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % ncol).fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 10 ** 18, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (name, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))


if __name__ == "__main__":
    main(sys.argv[1])
