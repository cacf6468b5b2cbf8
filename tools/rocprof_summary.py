#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs (kernel-trace stats and PMC counters) as text for profiles/."""
import sqlite3
import sys


def full_size_launches(db):
    """{kernel: (count, avg_us, max_us)} over the launches that run >= half as long as the kernel's longest one - the full-size
    launches of the benchmark; the one-point result checks and warm-ups of the same kernel are left out of this average."""
    for view in ("kernels", "rocpd_kernel_dispatch", "kernel_dispatch"):
        try:
            cols = [d[0] for d in db.execute(f"select * from {view} limit 1").description]
        except sqlite3.Error:
            continue
        name = next((c for c in ("name", "kernel_name", "kernel") if c in cols), None)
        if name and "duration" in cols:
            q = f"select {name}, duration from {view}"
        elif name and "start" in cols and "end" in cols:
            q = f"select {name}, end - start from {view}"
        else:
            continue
        per = {}
        for k, d in db.execute(q):
            if d is not None:
                per.setdefault(k, []).append(float(d) / 1e3)
        out = {}
        for k, ds in per.items():
            big = [d for d in ds if d >= 0.5 * max(ds)]
            out[k] = (len(big), sum(big) / len(big), max(ds))
        return out
    return {}


def kernel_stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    full = full_size_launches(db)
    out = [f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s} {'full-size launches: n':>22s} {'avg_us':>12s} {'max_us':>12s}"]
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("sv::", "")
        n, favg, fmax = full.get(name, (0, 0.0, 0.0))
        out.append(f"{short:70s} {calls:6d} {tot:12.1f} {avg:12.1f} {pct:7.2f} {n:22d} {favg:12.1f} {fmax:12.1f}")
    if not full:
        out.append("(no per-dispatch view found in this rocpd file: full-size columns are empty)")
    return "\n".join(out)


def schema(path):
    db = sqlite3.connect(path)
    out = []
    for typ, name in db.execute("select type, name from sqlite_master where type in ('table', 'view') order by name"):
        try:
            cols = [d[0] for d in db.execute(f"select * from '{name}' limit 1").description]
        except sqlite3.Error as e:
            cols = [str(e)]
        out.append(f"{typ} {name}: {', '.join(cols)}")
    return "\n".join(out)


def pmc(path):
    db = sqlite3.connect(path)
    cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
    rows = db.execute("select * from counters_collection").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    agg = {}
    for r in rows:
        key = (r[ix["kernel_name"]].split("(")[0].replace("sv::", ""), r[ix["counter_name"]])
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r[ix["value"]])
        a[2] = max(a[2], float(r[ix["value"]]))
    out = [f"{'kernel':60s} {'counter':20s} {'dispatches':>10s} {'sum':>16s} {'per_dispatch':>16s} {'largest dispatch':>18s}"]
    for (k, c), (n, s, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f"{k:60s} {c:20s} {n:10d} {s:16.1f} {s / n:16.1f} {mx:18.1f}")
    return "\n".join(out)


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    print(kernel_stats(path) if mode == "stats" else schema(path) if mode == "schema" else pmc(path))
