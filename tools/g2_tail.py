#!/usr/bin/env python3
"""The tail of a synchronous 2^16 G2 MSM over registered bases (fold + bit planes: the latency-bound part the round-5 review asked to shorten) under the
tuning in the environment: wall time per call and the HIP-event phases; meant to run under `rocprofv3 --kernel-trace --stats` for the per-kernel times
(tools/g2_tail.sh sweeps hex2 x fold_threads2 and prints one table).
  SNARKVM_HIP_TUNING=hex2=0,fold_threads2=128 python tools/g2_tail.py [tables=17] [window_bits=15] [lg=16]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from snarkvm_amd import _lib, synthetic  # noqa: E402
from snarkvm_amd.msm import RegisteredBasesG2  # noqa: E402


def main():
    tables = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    bits = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    lg = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    L = _lib.lib()
    n = 1 << lg
    bases = synthetic.g2_points(n)
    sc = synthetic.random_fr_integers(n, 4000 + lg)
    rb = RegisteredBasesG2(bases, tables=tables, window_bits=bits if tables != 16 else 0)
    first = rb.msm(sc)
    for _ in range(5):
        rb.msm(sc)
    reps = 30
    t0 = time.perf_counter()
    for _ in range(reps):
        got = rb.msm(sc)
    dt = (time.perf_counter() - t0) / reps
    from oracle import cpu as oracle  # (a check of this tool, after the timed loop; the raw Jacobian representative legitimately differs from call to call:
    # the scatter's atomics order the entries of a bucket differently every time)

    assert oracle.g2_to_affine(got).tobytes() == oracle.g2_to_affine(first).tobytes()
    L.snarkvm_hip_set_profiling(1)
    ph = {}
    for _ in range(5):
        rb.msm(sc)
        for i in range(L.snarkvm_hip_get_phase_count()):
            k = L.snarkvm_hip_get_phase_name(i).decode()
            ph[k] = ph.get(k, 0.0) + L.snarkvm_hip_get_phase_ms(i) / 5
    L.snarkvm_hip_set_profiling(0)
    rb.close()
    print(json.dumps({"tuning": os.environ.get("SNARKVM_HIP_TUNING", ""), "geometry": f"{tables} x {bits}", "lg": lg, "ms_per_sync_call_host_scalars": round(dt * 1e3, 4),
                      "phases_ms": {k: round(v, 4) for k, v in ph.items()}}))


if __name__ == "__main__":
    main()
