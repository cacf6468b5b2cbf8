#!/usr/bin/env python3
"""MSM over REGISTERED bases (precomputed tables resident in HBM) with HOST scalars: what a caller of the resident::Bases extension pays
per call - 32 B per pair over PCIe in scalar chunks whose upload hides behind the previous chunk's accumulation.
  SNARKVM_HIP_TUNING=taper=0 python tools/reg_host_scalars.py 22 24     (taper=0: one tail per scalar chunk, round 3)
Prints one markdown row; every result is checked (affine) against the unchunked MSM over device-resident scalars."""
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
    sizes = [int(a) for a in sys.argv[1:]] or [22, 24]
    L = _lib.lib()
    torch.cuda.set_device(0)
    nmax = 1 << max(sizes)
    buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
    sc = synthetic.random_fr_integers(nmax, 5)
    cells = []
    for lg in sizes:
        n = 1 << lg
        tb, bits = (12, 22) if lg >= 24 else (13, 20) if lg >= 21 else (16, 0)
        rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=tb, window_bits=bits)
        hs = sc[:n]
        first = rb.msm(hs)
        # the same sum with the scalars resident on the device: ONE unchunked MSM (its own sort, accumulate, tail) - another path to the same point
        dsc = torch.from_numpy(hs.view(np.int64).reshape(-1)).cuda()
        torch.cuda.synchronize()
        want = rb.msm(device_ptr=dsc.data_ptr(), npoints=n)
        assert _aff(first) == _aff(want), f"2^{lg}: host-scalar chunks and the device-scalar MSM disagree"
        del dsc
        reps = 4
        best, tot = 1e9, 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            r = rb.msm(hs)
            dt = time.perf_counter() - t0
            best, tot = min(best, dt), tot + dt
            assert _aff(r) == _aff(first)
        rb.close()
        cells.append(f"2^{lg} ({tb} x {bits or 16}): {tot / reps * 1e3:.2f} (best {best * 1e3:.2f})")
    print(f"| {os.environ.get('SNARKVM_HIP_TUNING', '(defaults)')} | " + " | ".join(cells) + " |")


def _aff(p):
    out = np.zeros(1, dtype=G1_AFFINE)
    _lib.check(_lib.lib().snarkvm_hip_g1_to_affine(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(p.ctypes.data), ctypes.c_size_t(1)))
    return out.tobytes()[:97]


if __name__ == "__main__":
    main()
