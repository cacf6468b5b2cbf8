#!/usr/bin/env python3
"""The "tables1" leg of the bench (93.6 ms per step in the driver's round-4 run against 37 ms everywhere else; not reproduced in round 5, not even with
round 4's library), taken apart.

The leg: a 2^24 MSM over registered bases WITHOUT precomputed tables (16 digit rows per scalar) right after the headline leg, whose
lanes hold workspaces sized for 12 digit rows.  Round 4 warmed ONE lane with a synchronous call and then timed a 2-instance batch:
lane 1 had to outgrow six buffers (2.8 GiB) inside the timed region, behind lane 0's running MSM.  This script times that exact shape on
the current library and reports what snarkvm_hip_alloc_stats saw in each call, then the shape the bench uses now (a warm-up batch of the
same shape over every lane), then hipFree / hipMalloc directly: a free issued while another stream runs kernels returns only when the
device is idle - the one thing in that timed region that can stall a device, now deferred to the end of the call."""
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

L = _lib.lib()
torch.cuda.set_device(0)
_lib.check(L.snarkvm_hip_set_device(0))
n = 1 << 24
bases = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
_lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(bases.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(n)))
sc = torch.from_numpy(synthetic.random_fr_integers(n, 1).view(np.int64)).cuda()
torch.cuda.synchronize()


def stats(reset=False):
    v = (ctypes.c_uint64 * 5)()
    L.snarkvm_hip_alloc_stats(v, 1 if reset else 0)
    return f"{v[0]} device allocations ({v[1] / 2**30:.2f} GiB), {v[4] / 1e3:.1f} ms inside them"


def timed(label, fn):
    _lib.check(L.snarkvm_hip_synchronize())
    stats(reset=True)
    t0 = time.perf_counter()
    fn()
    _lib.check(L.snarkvm_hip_synchronize())
    dt = (time.perf_counter() - t0) * 1e3
    print(f"| {label} | {dt:.1f} | {stats()} |")


print("| call | wall ms | workspace growth inside the call |")
print("|---|---|---|")
rb = RegisteredBases(device_ptr=bases.data_ptr(), npoints=n, tables=12, window_bits=22)
timed("headline geometry (12 x 22-bit tables), batch of 3: first use of three lanes", lambda: rb.msm_batch(device_ptrs=[sc.data_ptr()] * 3, npoints=[n] * 3))
timed("the same batch again", lambda: rb.msm_batch(device_ptrs=[sc.data_ptr()] * 3, npoints=[n] * 3))
rb1 = RegisteredBases(device_ptr=bases.data_ptr(), npoints=n, tables=1)
timed("no tables (16 digit rows): ONE synchronous call = round 4's warm-up (lane 0 grows)", lambda: rb1.msm(device_ptr=sc.data_ptr(), npoints=n))
timed("round 4's timed region: batch of 2 (lane 1 grows behind lane 0's running MSM)", lambda: rb1.msm_batch(device_ptrs=[sc.data_ptr()] * 2, npoints=[n] * 2))
timed("the same batch again (what the bench times now, after a warm-up batch over every lane)", lambda: rb1.msm_batch(device_ptrs=[sc.data_ptr()] * 2, npoints=[n] * 2))
timed("and again", lambda: rb1.msm_batch(device_ptrs=[sc.data_ptr()] * 2, npoints=[n] * 2))

# ---- what round 4's dev_buf::ensure did inside the timed region: hipFree of an outgrown block (+ hipMalloc of its successor), measured directly
hip = ctypes.CDLL("libamdhip64.so")


def raw_alloc(n_bytes):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n_bytes)) == 0
    return p


print("\n| six blocks of 768 MiB (what lane 1 outgrew) | ms |")
print("|---|---|")
blocks = [raw_alloc(768 << 20) for _ in range(6)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in blocks:
    assert hip.hipFree(b) == 0
print(f"| hipFree x 6, device idle | {(time.perf_counter() - t0) * 1e3:.2f} |")
blocks = [raw_alloc(768 << 20) for _ in range(6)]
a = torch.randn(8192, 8192, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    a = a @ a * 1e-4  # ~40 ms of queued kernels on torch's stream: the "other lane's running MSM"
t_q = time.perf_counter() - t0
t0 = time.perf_counter()
assert hip.hipFree(blocks[0]) == 0
t_first = time.perf_counter() - t0
t0 = time.perf_counter()
for b in blocks[1:]:
    assert hip.hipFree(b) == 0
t_rest = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"| enqueue of ~40 ms of kernels on another stream | {t_q * 1e3:.2f} |")
print(f"| first hipFree while those kernels run | {t_first * 1e3:.2f} |")
print(f"| the other five hipFree | {t_rest * 1e3:.2f} |")
t0 = time.perf_counter()
blocks = [raw_alloc(1 << 30) for _ in range(6)]
print(f"| hipMalloc x 6 of 1 GiB | {(time.perf_counter() - t0) * 1e3:.2f} |")
for b in blocks:
    hip.hipFree(b)
