#!/usr/bin/env python3
"""Dispatch-ordered trace of a window of a rocprofv3 rocpd database: start offset, duration and the idle gap to the previous kernel's end.
    python tools/rocpd_trace.py x.db k_init_state 2 400        # from the 3rd launch of k_init_state on, 400 kernels
Also prints, for the window, per-kernel totals and the sum of the gaps (time no kernel of the window was running)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    return re.sub(r"^void ", "", name)


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2]
    nth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    quiet = len(sys.argv) > 5 and sys.argv[5] == "summary"
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if nth >= len(idx):
        sys.exit("marker %r seen %d times" % (marker, len(idx)))
    win = rows[idx[nth]:idx[nth] + count]
    t0 = win[0][1]
    prev_end = t0
    gaps = 0
    agg = {}
    for n, s, e in win:
        g = s - prev_end
        if g > 0:
            gaps += g
        if not quiet:
            print("%10.1f us  %8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, g / 1e3, short(n)[:60]))
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1; a[1] += e - s
        prev_end = max(prev_end, e)
    span = prev_end - t0
    print("window: %d kernels, span %.1f us, idle gaps %.1f us (%.1f %%)" % (len(win), span / 1e3, gaps / 1e3, 100.0 * gaps / max(span, 1)))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-44s %5d x %8.1f us = %9.1f us" % (k[:44], a[0], a[1] / a[0] / 1e3, a[1] / 1e3))


if __name__ == "__main__":
    main()
