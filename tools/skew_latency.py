#!/usr/bin/env python3
"""Latency of a 2^16 / 2^18 MSM over registered bases for skewed scalar vectors (what witness polynomials look like): uniform,
half ones, 90 % small values, all equal.  The accumulate kernel is balanced by construction (fixed-length segments of the
sorted entries); the tail walks flattened partial-sum lists, so a bucket that receives most of the scalars costs a few more
additions in its row and column, not a reduce round."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from snarkvm_amd import _lib, synthetic
from snarkvm_amd.layout import G1_AFFINE
from snarkvm_amd.msm import RegisteredBases

L = _lib.lib(); torch.cuda.set_device(0)
nmax = 1 << 18
buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
_lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=nmax, tables=16)
rng = np.random.default_rng(7)
uni = synthetic.random_fr_integers(nmax, 5)
half_ones = uni.copy(); half_ones[rng.random(nmax) < 0.5] = [1, 0, 0, 0]
small = uni.copy(); m = rng.random(nmax) < 0.9; small[m] = 0; small[m, 0] = rng.integers(0, 256, int(m.sum()), dtype=np.uint64)
equal = np.tile(uni[:1], (nmax, 1))
print("| scalars | 2^16 ms | 2^18 ms |"); print("|---|---|---|")
for name, sc in [("uniform", uni), ("half ones", half_ones), ("90 % below 256", small), ("all equal", equal)]:
    d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda(); torch.cuda.synchronize()
    row = []
    for n in (1 << 16, 1 << 18):
        for _ in range(3): rb.msm(device_ptr=d.data_ptr(), npoints=n)
        t0 = time.perf_counter()
        for _ in range(10): rb.msm(device_ptr=d.data_ptr(), npoints=n)
        row.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"| {name} | {row[0]:.3f} | {row[1]:.3f} |", flush=True)
