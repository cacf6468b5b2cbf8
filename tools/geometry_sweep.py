#!/usr/bin/env python3
"""Which table geometry (J tables x b-bit windows, 254 <= J b <= 288) is fastest for a registered G1 MSM of 2^lg pairs?  Synchronous and
pipelined-batch times per geometry, one MI355X; every result is checked against the first geometry's (affine).  The rule the library
applies on its own (api.hip / bench.py): 17 x 15 below 2^18, 16 x 16 up to 2^20, 13 x 20 at 2^21 - 2^22, 12 x 22 from 2^23.
usage: python tools/geometry_sweep.py [lg ...]   (default 18 19 20 21 22)"""
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

GEOMETRIES = [(17, 15), (16, 16), (15, 17), (15, 18), (14, 19), (13, 20), (13, 21), (12, 22)]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [18, 19, 20, 21, 22]
    L = _lib.lib()
    torch.cuda.set_device(0)
    nmax = 1 << max(sizes)
    buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
    sc = synthetic.random_fr_integers(nmax, synthetic.SEED_MSM_LARGE)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    aff = np.zeros(1, dtype=G1_AFFINE)
    print("| lg n | tables x bits | sync ms | pipelined ms / instance | pairs/s (pipelined) |")
    print("|---|---|---|---|---|")
    for lg in sizes:
        n = 1 << lg
        ref = None
        for tables, bits in GEOMETRIES:
            try:
                rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=tables, window_bits=0 if bits == 16 else bits)
            except Exception as e:  # a geometry the registration refuses
                print(f"| {lg} | {tables} x {bits} | refused: {str(e)[:60]} | | |")
                continue
            r = rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
            _lib.check(L.snarkvm_hip_g1_to_affine(ctypes.c_void_p(aff.ctypes.data), ctypes.c_void_p(r.ctypes.data), ctypes.c_size_t(1)))
            if ref is None:
                ref = aff.tobytes()
            assert aff.tobytes() == ref, (lg, tables, bits)
            reps = 4 if lg >= 22 else 10
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
            rb.close()
            print(f"| {lg} | {tables} x {bits} | {sync_ms:.3f} | {pipe_ms:.3f} | {n / pipe_ms * 1e3:.3e} |", flush=True)


if __name__ == "__main__":
    main()
