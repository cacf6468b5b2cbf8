#!/usr/bin/env python3
"""Synchronous G1 MSM latency over registered bases (16 x 16-bit tables) at sizes between the powers of two: the accumulate
segment length fills whole rounds of one wave per SIMD, so the curve has no steps (profiles/r02_size_sweep.md)."""
import ctypes, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snarkvm_amd import _lib, synthetic
from snarkvm_amd.layout import G1_AFFINE
from snarkvm_amd.msm import RegisteredBases
L = _lib.lib(); torch.cuda.set_device(0)
nmax = 1 << 19
buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
_lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
sc = synthetic.random_fr_integers(nmax, 5)
d_sc = torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=nmax, tables=16)
for n in [256, 1024, 2048, 4095, 4096, 8192, 16384, 40000, 65536, 66000, 70000, 100000, 131072, 140000, 200000, 262144, 270000, 400000, 524288]:
    for _ in range(3): rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
    t0 = time.perf_counter()
    for _ in range(20): rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
    print(n, f"{(time.perf_counter()-t0)/20*1e3:.3f} ms", flush=True)
