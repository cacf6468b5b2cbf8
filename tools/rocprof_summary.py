#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs (kernel-trace stats and PMC counters) as text for profiles/."""
import sqlite3
import sys


def kernel_stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out = [f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}"]
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("sv::", "")
        out.append(f"{short:70s} {calls:6d} {tot:12.1f} {avg:12.1f} {pct:7.2f}")
    return "\n".join(out)


def pmc(path):
    db = sqlite3.connect(path)
    cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
    rows = db.execute("select * from counters_collection").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    agg = {}
    for r in rows:
        key = (r[ix["kernel_name"]].split("(")[0].replace("sv::", ""), r[ix["counter_name"]])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(r[ix["value"]])
    out = [f"{'kernel':60s} {'counter':14s} {'dispatches':>10s} {'sum':>16s} {'per_dispatch':>16s}"]
    for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f"{k:60s} {c:14s} {n:10d} {s:16.1f} {s / n:16.1f}")
    return "\n".join(out)


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    print(kernel_stats(path) if mode == "stats" else pmc(path))
