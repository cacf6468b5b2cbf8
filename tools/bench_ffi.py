#!/usr/bin/env python3
"""The reference's own FFI (`snarkvm_msm`, `snarkvm_ntt`: host buffers in, host buffers out) timed end to end on one MI355X -
the PCIe-inclusive figures DESIGN.md quotes next to the device-resident `bench.py` numbers.

  uncached   SNARKVM_HIP_BASE_CACHE=0: bases uploaded and converted on every call like algorithms/cuda/cuda/snarkvm.cu:262-275
             (big calls in point-range chunks whose upload overlaps the previous chunk's computation)
  cached     default: the host base range was seen before and lives in HBM with precomputed tables; only the scalars cross PCIe

Run once per mode (the base cache is configured by the environment at the first call):
  SNARKVM_HIP_BASE_CACHE=0 python tools/bench_ffi.py        # + SNARKVM_HIP_TRACE=1 for per-chunk host timestamps
  python tools/bench_ffi.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from snarkvm_amd import _lib, plugin, synthetic  # noqa: E402
from snarkvm_amd.layout import G1_AFFINE, NTTDirection, NTTInputOutputOrder, NTTType  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 20, 24]
    L = _lib.lib()
    torch.cuda.set_device(0)
    nmax = 1 << max(sizes)
    buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
    bases = buf.cpu().numpy().view(G1_AFFINE)
    del buf
    sc = synthetic.random_fr_integers(nmax, 5)
    x = synthetic.random_fr_integers(nmax, 6)
    mode = "uncached" if os.environ.get("SNARKVM_HIP_BASE_CACHE") == "0" else "cached (second call onwards)"
    print(f"mode: {mode}; chunk lg: {os.environ.get('SNARKVM_HIP_MSM_CHUNK_LG', '21 (default)')}")
    print("| lg n | snarkvm_msm ms (host bases + scalars) | pairs/s | snarkvm_ntt ms (host vector) | elements/s |")
    print("|---|---|---|---|---|")
    for lg in sizes:
        n = 1 << lg
        plugin.msm(bases[:n], sc[:n])
        plugin.msm(bases[:n], sc[:n])
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            plugin.msm(bases[:n], sc[:n])
        dt = (time.perf_counter() - t0) / reps
        y = x[:n].copy()
        plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Forward, NTTType.Standard)
        t0 = time.perf_counter()
        for i in range(reps):
            plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Forward if i % 2 == 0 else NTTDirection.Inverse, NTTType.Standard)
        dn = (time.perf_counter() - t0) / reps
        print(f"| {lg} | {dt * 1e3:.2f} | {n / dt:.3e} | {dn * 1e3:.2f} | {n / dn:.3e} |")


if __name__ == "__main__":
    main()
