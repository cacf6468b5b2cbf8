#!/usr/bin/env python3
"""Proof-sized commitment rounds over one registered SRS (17 x 15-bit tables): the 14 G1 MSMs of one transfer_private proof
(snarkvm_amd/proofs.py: 2^16 - 2^17 pairs each) issued
  all14     as ONE batched call (fused multi-instance launch sequence, runtime.hip.h::msm_batch_run)
  rounds    as the prover's rounds (1, 1, 2, 3, 4, 3 instances per call; sonic_pc/mod.rs:186-245)
  single    one synchronous call per instance
Device-resident scalars; every result is checked against the closed form (bases (i + 1) G).  Run twice for the A/B:
  python tools/bench_round.py                       # fused groups (default)
  SNARKVM_HIP_TUNING=fuse_batch=0 python tools/bench_round.py   # every instance through its own launch sequence on its own lane
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from snarkvm_amd import _lib, plugin, synthetic  # noqa: E402
from snarkvm_amd.layout import G1_AFFINE, G1_PROJECTIVE  # noqa: E402
from snarkvm_amd.msm import RegisteredBases  # noqa: E402


def main():
    L = _lib.lib()
    nR, nK = 1 << 16, 1 << 17
    sizes = [nR, nR, nR + 1, nR, nK - 1, nK - 1, nK - 1, nK - 2, nK, nR, nK, nK - 1, nR - 1, nK - 1]
    rounds = [1, 1, 2, 3, 4, 3]
    n = 2 * nK + 8
    buf = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(n)))
    tables, bits = int(os.environ.get("BENCH_ROUND_TABLES", "17")), int(os.environ.get("BENCH_ROUND_BITS", "15"))  # A/B of the geometry
    rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=tables, window_bits=0 if bits == 16 and tables == 16 else bits)
    host = [synthetic.random_fr_integers(k, 7700 + i) for i, k in enumerate(sizes)]
    dev = [torch.from_numpy(h.view(np.int64)).cuda() for h in host]
    torch.cuda.synchronize()
    ptrs = [d.data_ptr() for d in dev]
    gen = np.zeros(1, dtype=G1_AFFINE)
    gen["x"], gen["y"] = bench.G1_GEN_X, bench.G1_GEN_Y

    def to_affine(proj):
        proj = np.ascontiguousarray(proj, dtype=G1_PROJECTIVE).reshape(-1)
        out = np.zeros(proj.shape[0], dtype=G1_AFFINE)
        _lib.check(L.snarkvm_hip_g1_to_affine(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(proj.ctypes.data), ctypes.c_size_t(proj.shape[0])))
        return out

    want = []
    for h in host:
        k = bench.weighted_sum_mod_r(h, start=1)
        sc = np.array([[(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64)
        want.append(to_affine(plugin.msm(gen, sc)).tobytes())

    def all14():
        return rb.msm_batch(device_ptrs=ptrs, npoints=sizes)

    def by_rounds():
        out, i = [], 0
        for r in rounds:
            out.append(rb.msm_batch(device_ptrs=ptrs[i:i + r], npoints=sizes[i:i + r]))
            i += r
        return np.concatenate(out)

    def single():
        return np.concatenate([rb.msm(device_ptr=p, npoints=k) for p, k in zip(ptrs, sizes)])

    res = {"fuse_batch": os.environ.get("SNARKVM_HIP_TUNING", "(defaults)"), "tables_x_bits": f"{tables} x {bits}", "pairs": sum(sizes), "instances": len(sizes)}
    modes = os.environ.get("BENCH_ROUND_MODES", "all14,rounds,single").split(",")  # e.g. all14 alone under rocprofv3
    for name, fn in (("all14", all14), ("rounds", by_rounds), ("single", single)):
        if name not in modes:
            continue
        got = to_affine(fn())
        for i in range(len(sizes)):
            if got[i:i + 1].tobytes() != want[i]:
                raise SystemExit(f"bench_round: RESULT MISMATCH in {name}, instance {i}")
        fn()
        _lib.check(L.snarkvm_hip_synchronize())
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        res[name] = {"ms": dt * 1e3, "pairs_per_s": sum(sizes) / dt}
    # HIP-event phases of one fused all14 call (events on the launch stream of the lane that ran the group)
    L.snarkvm_hip_set_profiling(1)
    all14()
    res["all14_phase_ms"] = {L.snarkvm_hip_get_phase_name(i).decode(): round(L.snarkvm_hip_get_phase_ms(i), 4) for i in range(L.snarkvm_hip_get_phase_count())}
    rb.msm_batch(device_ptrs=ptrs[:1], npoints=sizes[:1])
    res["one_instance_batch_phase_ms"] = {L.snarkvm_hip_get_phase_name(i).decode(): round(L.snarkvm_hip_get_phase_ms(i), 4) for i in range(L.snarkvm_hip_get_phase_count())}
    L.snarkvm_hip_set_profiling(0)
    print(json.dumps(res))
    rb.close()


if __name__ == "__main__":
    main()
