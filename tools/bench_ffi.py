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
    print(f"mode: {mode}; tuning: {os.environ.get('SNARKVM_HIP_TUNING', '(defaults)')}")
    print("| lg n | snarkvm_msm ms (host bases + scalars) | pairs/s | snarkvm_ntt ms (host vector) | elements/s |")
    print("|---|---|---|---|---|")
    rows_poly = []
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
        rows_poly.append((lg, n))
    # snarkvm_polymul: PolyMultiplier::multiply of k coefficient vectors of n / k elements each on the 2^lg domain (host buffers in,
    # the full-domain product out); the upload of operand i + 1 overlaps the transform of operand i (polynomial.cuh:136-242)
    print()
    print("| lg domain | operands | snarkvm_polymul ms | PCIe bytes (in + out) | effective GB/s |")
    print("|---|---|---|---|---|")
    for lg, n in rows_poly:
        for k in (2, 4):
            polys = [np.ascontiguousarray(x[i * (n // k): (i + 1) * (n // k)]) for i in range(k)]
            out = np.zeros((n, 4), dtype=np.uint64)
            out[:] = 0  # touched pages, like Rust's vec![zero; domain] (lib.rs:126-127); calloc'ed pages fault during the download
            pp = (ctypes.c_void_p * k)(*[p.ctypes.data for p in polys])
            pl = (ctypes.c_size_t * k)(*[p.shape[0] for p in polys])
            call = lambda: _lib.check(L.snarkvm_polymul(ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(k), pp, pl, ctypes.c_size_t(0), None, None, ctypes.c_uint32(lg)))  # noqa: E731
            call()
            t0 = time.perf_counter()
            for _ in range(3):
                call()
            dp = (time.perf_counter() - t0) / 3
            moved = 32 * (n + n)
            print(f"| {lg} | {k} x 2^{lg} / {k} | {dp * 1e3:.2f} | {moved / 1e6:.0f} MB | {moved / dp / 1e9:.1f} |")


if __name__ == "__main__":
    main()
