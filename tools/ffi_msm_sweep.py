#!/usr/bin/env python3
"""`snarkvm_msm` over host buffers (stateless: SNARKVM_HIP_BASE_CACHE=0) timed end to end at a few sizes for the tuning in the environment:
  SNARKVM_HIP_BASE_CACHE=0 SNARKVM_HIP_TUNING=ramp=0 python tools/ffi_msm_sweep.py 20 22 24
Prints one markdown row; every size's result is compared with the first call's (affine)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from snarkvm_amd import _lib, plugin, synthetic  # noqa: E402
from snarkvm_amd.layout import G1_AFFINE  # noqa: E402


def affine(p):
    out = np.zeros(1, dtype=G1_AFFINE)
    _lib.check(_lib.lib().snarkvm_hip_g1_to_affine(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(p.ctypes.data), ctypes.c_size_t(1)))
    return out.tobytes()[:97]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [20, 22, 24]
    L = _lib.lib()
    torch.cuda.set_device(0)
    nmax = 1 << max(sizes)
    buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
    bases = buf.cpu().numpy().view(G1_AFFINE)
    del buf
    sc = synthetic.random_fr_integers(nmax, 5)
    cells = []
    for lg in sizes:
        n = 1 << lg
        first = affine(plugin.msm(bases[:n], sc[:n]))
        plugin.msm(bases[:n], sc[:n])
        reps = 5 if lg < 24 else 4
        best, tot = 1e9, 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            r = plugin.msm(bases[:n], sc[:n])
            dt = time.perf_counter() - t0
            best, tot = min(best, dt), tot + dt
            assert affine(r) == first, "result changed between calls"
        cells.append(f"2^{lg}: {tot / reps * 1e3:.2f} (best {best * 1e3:.2f})")
    print(f"| {os.environ.get('SNARKVM_HIP_TUNING', '(defaults)')} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
