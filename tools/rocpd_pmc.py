#!/usr/bin/env python3
"""Per-kernel PMC summary of a rocprofv3 rocpd database (run with --kernel-trace --pmc ...), CSV on stdout:
    python tools/rocpd_pmc.py x_results.db > profiles/rNN_x_pmc.csv
Counter values are summed over the counter's instances (XCDs / SEs / SIMDs) per dispatch, then averaged over the dispatches of a kernel."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    return re.sub(r"^void ", "", name)


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select dispatch_id, kernel_name, counter_name, value, start, end from counters_collection").fetchall()
    per_dispatch = defaultdict(float)
    meta = {}
    for did, kname, cname, val, s, e in rows:
        per_dispatch[(did, cname)] += val
        meta[did] = (short(kname), e - s)
    agg = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    seen = set()
    for (did, cname), v in per_dispatch.items():
        k, d = meta[did]
        agg[k][cname].append(v)
        if did not in seen:
            durs[k].append(d); seen.add(did)
    counters = sorted({c for k in agg for c in agg[k]})
    print("kernel,dispatches,avg_ns_under_pmc," + ",".join("avg_%s,max_%s" % (c, c) for c in counters))
    for k in sorted(agg, key=lambda k_: -sum(durs[k_])):
        vals = []
        for c in counters:
            x = agg[k].get(c, [0.0])
            vals += ["%.1f" % (sum(x) / len(x)), "%.1f" % max(x)]
        print('"%s",%d,%.1f,%s' % (k, len(durs[k]), sum(durs[k]) / len(durs[k]), ",".join(vals)))


if __name__ == "__main__":
    main(sys.argv[1])
