#!/usr/bin/env python3
"""G2 variable-base MSM (BASELINE.json configs[4]: 2^16 per proof instance) on one MI355X through the host-pointer ABI
(`snarkvm_hip_msm_g2`: bases and scalars cross PCIe on every call), per-phase kernel times from the HIP-event profiler,
next to the CPU oracle's `standard::msm` (the path G2 takes in the reference).  Bases: 1024 distinct multiples of the G2
generator tiled to n (the reference's benches tile a small random set the same way)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import cpu as oracle  # noqa: E402  (base generation + CPU baseline leg)
from snarkvm_amd import _lib, synthetic  # noqa: E402
from snarkvm_amd.msm import RegisteredBasesG2, msm_g2  # noqa: E402


def main():
    L = _lib.lib()
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "constants.json")))["g2"]
    gen = np.zeros(1, dtype=oracle.G2_AFFINE)
    gen["x"] = golden["G2_GENERATOR_X_C0_MONT"] + golden["G2_GENERATOR_X_C1_MONT"]
    gen["y"] = golden["G2_GENERATOR_Y_C0_MONT"] + golden["G2_GENERATOR_Y_C1_MONT"]
    proj = np.zeros(1024, dtype=oracle.G2_PROJECTIVE)
    for i in range(1024):
        s = np.array([3 * i + 1, 0, 0, 0], dtype=np.uint64)
        proj[i] = oracle.g2_mul(gen, s)[0]
    distinct = oracle.g2_to_affine(proj)
    oracle.set_threads(min(64, oracle.max_threads()))
    print("| lg n | one-shot GPU ms (host buffers, PCIe incl.) | pairs/s | registered (16 tables) ms | pairs/s | kernel phases ms | CPU standard::msm pairs/s |")
    print("|---|---|---|---|---|---|---|")
    for lg in (12, 16, 18):
        n = 1 << lg
        bases = np.tile(distinct, n // 1024)
        sc = synthetic.random_fr_integers(n, 4000 + lg)
        got = msm_g2(bases, sc)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            msm_g2(bases, sc)
        dt = (time.perf_counter() - t0) / reps
        rb = RegisteredBasesG2(bases, tables=16)
        got_r = rb.msm(sc)
        t0 = time.perf_counter()
        for _ in range(reps):
            rb.msm(sc)
        dtr = (time.perf_counter() - t0) / reps
        assert oracle.g2_to_affine(got_r).tobytes() == oracle.g2_to_affine(got).tobytes()
        L.snarkvm_hip_set_profiling(1)
        rb.msm(sc)
        phr = {L.snarkvm_hip_get_phase_name(i).decode(): round(L.snarkvm_hip_get_phase_ms(i), 3) for i in range(L.snarkvm_hip_get_phase_count())}
        msm_g2(bases, sc)
        ph = {L.snarkvm_hip_get_phase_name(i).decode(): round(L.snarkvm_hip_get_phase_ms(i), 3) for i in range(L.snarkvm_hip_get_phase_count())}
        L.snarkvm_hip_set_profiling(0)
        rb.close()
        ph = {"registered": phr, "one_shot": ph}
        cpu = ""
        if lg <= 16:
            t0 = time.perf_counter()
            want = oracle.g2_msm(bases, sc)
            cpu = f"{n / (time.perf_counter() - t0):.3e}"
            assert oracle.g2_to_affine(got).tobytes() == oracle.g2_to_affine(want).tobytes()
        print(f"| {lg} | {dt * 1e3:.3f} | {n / dt:.3e} | {dtr * 1e3:.3f} | {n / dtr:.3e} | {ph} | {cpu} |")


if __name__ == "__main__":
    main()
