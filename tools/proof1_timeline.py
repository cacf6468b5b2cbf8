#!/usr/bin/env python3
"""Timeline of ONE proof proved by one caller thread (BASELINE.json configs[3]): where the wall time of `replay_single` goes.

  run      (under rocprofv3 --kernel-trace):  python tools/proof1_timeline.py run <marks.json> [--sync-msm | --await] [--proofs N]
           proves N proofs one at a time (5 ms pause between them) and writes the host-side step marks (three clocks each)
  report   python tools/proof1_timeline.py report <rocpd.db> <marks.json>  -> markdown on stdout:
           per proof: host begin -> first kernel -> last kernel -> results on the host, GPU-busy share (union of the kernel intervals of
           all streams / the proof's wall time); for the median proof: every host mark and, per stream, what ran when.
"""
import json
import os
import sqlite3
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def run(path, sync_msm, count, awaited=False):
    import torch

    from snarkvm_amd import _lib, proofs

    torch.cuda.set_device(0)
    L = _lib.lib()
    _lib.check(L.snarkvm_hip_set_device(0))
    shape = proofs.ProofShape()
    keys = proofs.ProverKeys(shape, tables=17, window_bits=15)
    ws = proofs.SingleProofWorkspace(keys)
    for s in range(4):
        proofs.replay_single(ws, s, None, not sync_msm, None, awaited, awaited)
    out = []
    for s in range(count):
        time.sleep(0.005)
        marks = []
        proofs.replay_single(ws, s, None, not sync_msm, marks, awaited, awaited)  # awaited: the transcript order (rounds collected, in-stream)
        out.append(marks)
    json.dump({"sync_msm": sync_msm, "proofs": out}, open(path, "w"))
    keys.close()


def union_ns(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def short(name):
    n = name.split("(")[0].replace("sv::", "").replace("void ", "")
    return n[:60]


def report(db_path, marks_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, stream_id, start, end from kernels order by start").fetchall()
    marks = json.load(open(marks_path))
    proofs_m = marks["proofs"]
    # which host clock does the trace use?  the one that puts the first kernel of every proof shortly after its "begin" mark
    best = None
    for ci in range(4):
        ok, lag = 0, []
        for pm in proofs_m:
            b, e = pm[0][1][ci], pm[-1][1][ci]
            inside = [r for r in rows if b <= r[2] <= e]
            if inside:
                ok += 1
                lag.append(inside[0][2] - b)
        if ok == len(proofs_m) and (best is None or sum(lag) < best[1]):
            best = (ci, sum(lag))
    if best is None:
        print("no host clock lines up with the trace's timestamps (clock domains differ); kernel clusters only")
        return
    ci = best[0]
    cname = ["CLOCK_MONOTONIC", "CLOCK_MONOTONIC_RAW", "CLOCK_BOOTTIME", "CLOCK_REALTIME"][ci]
    print(f"mode: {'synchronous commitments' if marks['sync_msm'] else 'SNARKVM_HIP_SCOPE_ASYNC_MSM'}; trace timestamps = {cname}; {len(proofs_m)} proofs, one at a time\n")
    print("| proof | wall ms (begin -> results on host) | host enqueue ms | first kernel +us | last kernel end ms | kernels | sum of kernel ms | GPU busy (union) ms | busy share | streams |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    per = []
    for i, pm in enumerate(proofs_m):
        b, e = pm[0][1][ci], pm[-1][1][ci]
        enq = pm[-2][1][ci]
        ks = [r for r in rows if b <= r[2] <= e]
        if not ks:
            continue
        busy = union_ns([(r[2], r[3]) for r in ks])
        wall = e - b
        per.append((wall, i, ks, busy))
        print(f"| {i} | {wall / 1e6:.3f} | {(enq - b) / 1e6:.3f} | {(ks[0][2] - b) / 1e3:.0f} | {(max(r[3] for r in ks) - b) / 1e6:.3f} | {len(ks)} | "
              f"{sum(r[3] - r[2] for r in ks) / 1e6:.3f} | {busy / 1e6:.3f} | {busy / wall:.2f} | {len(set(r[1] for r in ks))} |")
    per.sort()
    wall, i, ks, busy = per[len(per) // 2]
    walls = [p[0] for p in per]
    print(f"\nmedian wall {walls[len(walls) // 2] / 1e6:.3f} ms, min {walls[0] / 1e6:.3f}, max {walls[-1] / 1e6:.3f}; GPU busy share of the median proof {busy / wall:.2f}\n")
    pm = proofs_m[i]
    b = pm[0][1][ci]
    print(f"### the median proof ({i}): host marks\n")
    print("| t (ms) | host |")
    print("|---|---|")
    for label, clocks in pm:
        print(f"| {(clocks[ci] - b) / 1e6:.3f} | {label} |")
    print("\n### the same proof: per stream (start of the first, end of the last kernel, busy time, launches; then the longest kernels)\n")
    streams = {}
    for r in ks:
        streams.setdefault(r[1], []).append(r)
    print("| stream | first kernel start ms | last kernel end ms | busy ms | launches | what (summed ms) |")
    print("|---|---|---|---|---|---|")
    for sid, rs in sorted(streams.items(), key=lambda kv: kv[1][0][2]):
        byname = {}
        for r in rs:
            byname[short(r[0])] = byname.get(short(r[0]), 0) + (r[3] - r[2])
        top = ", ".join(f"{k} {v / 1e6:.3f}" for k, v in sorted(byname.items(), key=lambda kv: -kv[1])[:6])
        print(f"| {sid} | {(rs[0][2] - b) / 1e6:.3f} | {(max(r[3] for r in rs) - b) / 1e6:.3f} | {union_ns([(r[2], r[3]) for r in rs]) / 1e6:.3f} | {len(rs)} | {top} |")
    print("\n### the same proof: every kernel (launches, summed ms, share of the summed kernel time)\n")
    byname = {}
    for r in ks:
        e = byname.setdefault(short(r[0]), [0, 0])
        e[0] += 1
        e[1] += r[3] - r[2]
    tot = sum(v[1] for v in byname.values())
    print("| kernel | launches | ms | share |")
    print("|---|---|---|---|")
    for k, v in sorted(byname.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {v[0]} | {v[1] / 1e6:.3f} | {v[1] / tot:.3f} |")
    print(f"| all | {len(ks)} | {tot / 1e6:.3f} | 1 |")
    print("\n### the same proof: every accumulate launch (one per commitment round / MSM) and the gaps of the transform stream\n")
    print("| kernel | stream | start ms | end ms | ms |")
    print("|---|---|---|---|---|")
    for r in ks:
        if "accumulate" in r[0]:
            print(f"| {short(r[0])} | {r[1]} | {(r[2] - b) / 1e6:.3f} | {(r[3] - b) / 1e6:.3f} | {(r[3] - r[2]) / 1e6:.3f} |")
    main_sid = max(streams.items(), key=lambda kv: sum(1 for r in kv[1] if "ntt_pass" in r[0]))[0]
    rs = streams[main_sid]
    gaps = sorted(((rs[k + 1][2] - rs[k][3], rs[k][3], short(rs[k][0]), short(rs[k + 1][0])) for k in range(len(rs) - 1)), reverse=True)[:8]
    print(f"\nlargest idle gaps of the transform stream ({main_sid}):\n")
    print("| gap us | at ms | after | before |")
    print("|---|---|---|---|")
    for g, at, a, bb in gaps:
        print(f"| {g / 1e3:.0f} | {(at - b) / 1e6:.3f} | {a} | {bb} |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        n = int(sys.argv[sys.argv.index("--proofs") + 1]) if "--proofs" in sys.argv else 16
        run(sys.argv[2], "--sync-msm" in sys.argv, n, "--await" in sys.argv)
    else:
        report(sys.argv[2], sys.argv[3])
