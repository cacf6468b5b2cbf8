#!/usr/bin/env python3
"""Per-phase HIP-event times of one synchronous G1 MSM over registered bases at small and large sizes (the latency-bound
regime a Varuna proof lives in: 2^14 .. 2^18), plus wall time per call.  Prints a markdown table."""
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
    sizes = [int(a) for a in sys.argv[1:]] or [14, 16, 17, 18, 20]
    L = _lib.lib()
    torch.cuda.set_device(0)
    nmax = 1 << max(sizes)
    buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
    sc = synthetic.random_fr_integers(nmax, synthetic.SEED_MSM_LARGE)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    for lg in sizes:
        n = 1 << lg
        bits = 22 if lg >= 23 else 20 if lg >= 21 else 16 if lg >= 18 else 15  # the base cache's geometry rule (api.hip)
        tables = 16 if bits == 16 else -(-254 // bits)
        if os.environ.get("PHASE_TABLES"):  # experiments: e.g. PHASE_TABLES=17 -> 17 x 15-bit tables
            tables = int(os.environ["PHASE_TABLES"])
            bits = int(os.environ.get("PHASE_BITS", 256 // tables))
        rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=tables, window_bits=0 if bits * tables == 256 else bits)
        for _ in range(3):
            rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
        wall = (time.perf_counter() - t0) / reps * 1e3
        L.snarkvm_hip_set_profiling(1)
        ph = {}
        order = []
        for _ in range(5):
            rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
            for i in range(L.snarkvm_hip_get_phase_count()):
                name = L.snarkvm_hip_get_phase_name(i).decode()
                if name not in ph:
                    order.append(name)
                ph[name] = ph.get(name, 0.0) + L.snarkvm_hip_get_phase_ms(i) / 5
        L.snarkvm_hip_set_profiling(0)
        rb.close()
        print(f"### 2^{lg} ({tables} x {bits}-bit tables): {wall:.3f} ms per synchronous call; phases sum {sum(ph.values()):.3f} ms")
        print("| " + " | ".join(order) + " |")
        print("|" + "---|" * len(order))
        print("| " + " | ".join(f"{ph[k]:.3f}" for k in order) + " |")
        print()


if __name__ == "__main__":
    main()
