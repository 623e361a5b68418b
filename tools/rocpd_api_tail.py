#!/usr/bin/env python3
"""HIP API calls of the last bench step of a rocprofv3 --runtime-trace --kernel-trace database (host-side view of a step).

  python tools/rocpd_api_tail.py <db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print("# tables/views:", [t for t in tabs if "rocpd_" not in t][:40])
k = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(k) if "k_index_insert" in r[0]]
a, b = k[idx[-2]][1], k[idx[-1]][1]
for view in ("regions", "regions_and_samples", "hip_api", "api"):
    if view in tabs:
        cols = [d[1] for d in c.execute(f"pragma table_info({view})")]
        print("# using", view, cols)
        ncol = "name" if "name" in cols else cols[0]
        rows = c.execute(f"select {ncol}, start, end from {view} where start >= ? and start < ? order by start", (a - 200000, b)).fetchall()
        for n, s, e in rows:
            print(f"{(s - a) / 1e3:9.1f} | {(e - s) / 1e3:7.1f} | {n}")
        break
