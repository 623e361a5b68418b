#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (this image's rocprofv3 writes SQLite, not CSV).

  python tools/rocpd_summary.py gpurun_out/prof/bench_results.db [--timeline] [--pmc]

default     per-kernel calls / total / average duration (the `--stats` view)
--timeline  the kernels of the last bench step in launch order with the idle gaps between them
--pmc       per-kernel sums of the collected counters (a `--pmc` run)"""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--pmc-seq", default=None, metavar="KERNEL",
                    help="with --pmc: every dispatch of the kernels whose name contains KERNEL, in dispatch order, one line each")
    ap.add_argument("--step-marker", default="k_index_", help="kernel that starts a bench step")
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    if args.pmc:
        rows = c.execute("select * from counters_collection limit 1")
        cols = [d[0] for d in rows.description]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        ncol = "counter_name" if "counter_name" in cols else "pmc_name"
        vcol = "value" if "value" in cols else "counter_value"
        if args.pmc_seq:
            dcol = next((x for x in ("dispatch_id", "dispatch_index", "id") if x in cols), None)
            per = collections.OrderedDict()
            q = f"select {dcol}, {kcol}, {ncol}, {vcol} from counters_collection order by {dcol}" if dcol else None
            if q is None:
                print("# no dispatch id column in counters_collection:", cols)
                return
            for d, k, n, v in c.execute(q):
                if args.pmc_seq not in k:
                    continue
                e = per.setdefault(d, {"kernel": k.split("(")[0].replace("ipcfp::", "")})
                e[n] = e.get(n, 0.0) + float(v)
            names = sorted({n for e in per.values() for n in e if n != "kernel"})
            print("# dispatch | kernel | " + " | ".join(names))
            for d, e in per.items():
                print(f"{d} | {e['kernel']} | " + " | ".join(f"{e.get(n, 0):.6g}" for n in names))
            return
        agg = collections.defaultdict(lambda: [0, 0.0])
        for k, n, v in c.execute(f"select {kcol}, {ncol}, {vcol} from counters_collection"):
            a = agg[(k.split("(")[0], n)]
            a[0] += 1
            a[1] += float(v)
        print("# kernel | counter | samples | sum | per-dispatch-sample mean")
        for (k, n), (cnt, tot) in sorted(agg.items()):
            print(f"{k} | {n} | {cnt} | {tot:.6g} | {tot / cnt:.6g}")
        return
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    if args.timeline:
        idx = [i for i, r in enumerate(rows) if args.step_marker in r[0]]
        a, b = idx[-2], idx[-1]
        t0, prev = rows[a][1], None
        busy = gaps = 0.0
        print("# start_us | dur_us | gap_before_us | kernel   (one bench step, launch order)")
        for n, s, e in rows[a:b]:
            gap = (s - prev) / 1e3 if prev else 0.0
            busy += (e - s) / 1e3
            gaps += max(gap, 0.0)
            print(f"{(s - t0) / 1e3:9.1f} | {(e - s) / 1e3:8.1f} | {gap:7.1f} | {n.split('(')[0].replace('ipcfp::', '').replace('void ', '')}")
            prev = e
        print(f"# kernels {b - a}, busy {busy:.1f} us, idle gaps {gaps:.1f} us, span {(rows[b][1] - t0) / 1e3:.1f} us")
        return
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s, e in rows:
        a = agg[n]
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values()) or 1
    print("# name | calls | total_us | avg_us | pct")
    for n, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:110]} | {cnt} | {ns / 1e3:.1f} | {ns / 1e3 / cnt:.2f} | {100.0 * ns / total:.2f}")


if __name__ == "__main__":
    main()
