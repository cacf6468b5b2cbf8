#!/usr/bin/env python3
"""Timing of the group-element iFFT (`UniversalParams::lagrange_basis`, polycommit/kzg10/data_structures.rs:68-72) through the C ABI at 2^10 ... 2^16, four lanes per
butterfly (default) against one lane per butterfly (SNARKVM_HIP_TUNING=group_quad=0, a child process per setting): python tools/group_ntt_timing.py [out.md]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np

    from snarkvm_amd import _lib, group
    from snarkvm_amd.layout import G1_AFFINE, G1_PROJECTIVE
    from snarkvm_amd.devmem import HipMem

    L = _lib.lib()
    nmax = 1 << 16
    d = HipMem(nmax * G1_AFFINE.itemsize)
    import ctypes

    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(d.ptr), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
    aff = d.download().view(G1_AFFINE)
    out = {}
    for lg in (10, 12, 14, 16):
        n = 1 << lg
        a = aff[:n]
        proj = np.zeros(n, dtype=G1_PROJECTIVE)
        proj["x"], proj["y"] = a["x"], a["y"]
        proj["z"] = np.array([202099033278250856, 5854854902718660529, 11492539364873682930, 8885205928937022213, 5545221690922665192, 39800542322357402], dtype=np.uint64)
        group.group_ntt(proj[:64], inverse=True)
        best = 1e9
        for _ in range(2 if lg >= 16 else 3):
            t0 = time.perf_counter()
            r = group.group_ntt(proj, inverse=True)
            best = min(best, time.perf_counter() - t0)
        out[str(lg)] = best * 1e3
    print(json.dumps(out))


def main():
    if os.environ.get("GROUP_NTT_CHILD"):
        return child()
    rows = {}
    for name, tune in (("four lanes per butterfly (group_quad=1)", ""), ("one lane per butterfly (group_quad=0, round 5)", "group_quad=0")):
        env = dict(os.environ, GROUP_NTT_CHILD="1", SNARKVM_HIP_TUNING=tune)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=1800)
        line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
        rows[name] = json.loads(line) if line else {"error": r.stderr[-500:]}
    md = ["| group iFFT through the C ABI (host buffers in and out), ms | 2^10 | 2^12 | 2^14 | 2^16 |", "|---|---|---|---|---|"]
    for name, v in rows.items():
        md.append(f"| {name} | " + " | ".join(f"{v.get(str(lg), float('nan')):.1f}" if isinstance(v.get(str(lg)), float) else "?" for lg in (10, 12, 14, 16)) + " |")
    text = "\n".join(md)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
