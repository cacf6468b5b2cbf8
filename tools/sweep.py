#!/usr/bin/env python3
"""Size sweep on one MI355X: G1 MSM (registered bases with precomputed tables: 17 x 15-bit below 2^18, 16 x 16-bit below 2^21, 13 x 20-bit at
2^22, 12 x 22-bit at 2^24 - the rule bench.py applies; synchronous and pipelined batch of 8) and Fr NTT (device resident,
NN forward) for 2^14 .. 2^24.  Prints a markdown table (committed under profiles/)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from snarkvm_amd import _lib, synthetic  # noqa: E402
from snarkvm_amd.layout import G1_AFFINE  # noqa: E402
from snarkvm_amd.msm import RegisteredBases  # noqa: E402


def main():
    L = _lib.lib()
    torch.cuda.set_device(0)
    nmax = 1 << 24
    buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
    sc = synthetic.random_fr_integers(nmax, synthetic.SEED_MSM_LARGE)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    d_x = torch.from_numpy(synthetic.random_fr_integers(nmax, synthetic.SEED_NTT).view(np.int64)).cuda()
    torch.cuda.synchronize()
    print("| lg n | base tables x bits | MSM sync ms | MSM pipelined ms/instance | MSM pairs/s (pipelined) | NTT ms | NTT elements/s |")
    print("|---|---|---|---|---|---|---|")
    for lg in range(14, 25, 2):
        n = 1 << lg
        reps = 3 if lg >= 22 else 8
        bits = 22 if lg >= 23 else 20 if lg >= 21 else 16 if lg >= 18 else 15  # the base cache's geometry rule (api.hip)
        tables = 16 if bits == 16 else -(-254 // bits)
        rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=tables, window_bits=0 if bits == 16 else bits)
        rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
        t0 = time.perf_counter()
        for _ in range(reps):
            rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
        sync_ms = (time.perf_counter() - t0) / reps * 1e3
        lanes = L.snarkvm_hip_batch_lanes(ctypes.c_size_t(n))
        rb.msm_batch(device_ptrs=[d_sc.data_ptr()] * lanes, npoints=[n] * lanes)
        k = 12
        t0 = time.perf_counter()
        rb.msm_batch(device_ptrs=[d_sc.data_ptr()] * k, npoints=[n] * k)
        pipe_ms = (time.perf_counter() - t0) / k * 1e3
        _lib.check(L.snarkvm_hip_ntt_device(ctypes.c_void_p(d_x.data_ptr()), ctypes.c_uint32(lg), 0, 0, 0))
        nrep = 20
        t0 = time.perf_counter()
        for i in range(nrep):
            _lib.check(L.snarkvm_hip_ntt_device(ctypes.c_void_p(d_x.data_ptr()), ctypes.c_uint32(lg), 0, i & 1, 0))
        ntt_ms = (time.perf_counter() - t0) / nrep * 1e3
        rb.close()
        print(f"| {lg} | {tables} x {bits} | {sync_ms:.3f} | {pipe_ms:.3f} | {n / pipe_ms * 1e3:.3e} | {ntt_ms:.4f} | {n / ntt_ms * 1e3:.3e} |")


if __name__ == "__main__":
    main()
