#!/usr/bin/env python3
"""The reference's own FFI (`snarkvm_msm`, `snarkvm_ntt`: host buffers in, host buffers out, bases re-uploaded and
converted on every call like algorithms/cuda/cuda/snarkvm.cu:262-275) timed end to end on one MI355X - the PCIe-inclusive
figures DESIGN.md quotes next to the device-resident `bench.py` numbers.  Run with SNARKVM_HIP_BASE_CACHE=16 to time the same
unmodified calls with the optional base cache (first call registers, later calls hit)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from snarkvm_amd import _lib, plugin, synthetic  # noqa: E402
from snarkvm_amd.layout import G1_AFFINE, NTTDirection, NTTInputOutputOrder, NTTType  # noqa: E402
import ctypes  # noqa: E402


def main():
    L = _lib.lib()
    torch.cuda.set_device(0)
    nmax = 1 << 24
    buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
    bases = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=G1_AFFINE)
    del buf
    sc = synthetic.random_fr_integers(nmax, 5)
    x = synthetic.random_fr_integers(nmax, 6)
    print("| lg n | snarkvm_msm ms (host bases + scalars) | pairs/s | phases ms | snarkvm_ntt ms (host vector) | elements/s |")
    print("|---|---|---|---|---|---|")
    for lg in (16, 20, 24):
        n = 1 << lg
        plugin.msm(bases[:n], sc[:n])
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            plugin.msm(bases[:n], sc[:n])
        dt = (time.perf_counter() - t0) / reps
        L.snarkvm_hip_set_profiling(1)
        plugin.msm(bases[:n], sc[:n])
        ph = {L.snarkvm_hip_get_phase_name(i).decode(): round(L.snarkvm_hip_get_phase_ms(i), 2) for i in range(L.snarkvm_hip_get_phase_count())}
        L.snarkvm_hip_set_profiling(0)
        y = x[:n].copy()
        plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Forward, NTTType.Standard)
        t0 = time.perf_counter()
        for i in range(reps):
            plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Forward if i % 2 == 0 else NTTDirection.Inverse, NTTType.Standard)
        dn = (time.perf_counter() - t0) / reps
        print(f"| {lg} | {dt * 1e3:.2f} | {n / dt:.3e} | {ph} | {dn * 1e3:.2f} | {n / dn:.3e} |")


if __name__ == "__main__":
    main()
