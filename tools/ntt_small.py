#!/usr/bin/env python3
"""Device-resident Fr NTTs at the domain sizes of a proof (2^12 .. 2^20): microseconds per transform over back-to-back calls on one
lane, for one value of the tuning key ntt_min_tiles (the rule that narrows the tiles of small transforms so that the launch covers
the chip; 1 = the [2^a x 8] tiles of round 2).  usage: SNARKVM_HIP_TUNING=ntt_min_tiles=k python tools/ntt_small.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from snarkvm_amd import _lib, synthetic  # noqa: E402


def main():
    L = _lib.lib()
    torch.cuda.set_device(0)
    print(f"SNARKVM_HIP_TUNING={os.environ.get('SNARKVM_HIP_TUNING', '(defaults)')}")
    print("| lg n | us per forward NTT (200 back-to-back device calls) | elements/s |")
    print("|---|---|---|")
    for lg in range(12, 21):
        n = 1 << lg
        x = torch.from_numpy(synthetic.random_fr_integers(n, 77 + lg).view("int64")).cuda()
        call = lambda d: _lib.check(L.snarkvm_hip_ntt_device(ctypes.c_void_p(x.data_ptr()), ctypes.c_uint32(lg), 0, d, 0))  # noqa: E731
        for _ in range(5):
            call(0)
        _lib.check(L.snarkvm_hip_synchronize())
        reps = 200
        t0 = time.perf_counter()
        for i in range(reps):
            call(i & 1)
        _lib.check(L.snarkvm_hip_synchronize())
        dt = (time.perf_counter() - t0) / reps
        print(f"| {lg} | {dt * 1e6:.1f} | {n / dt:.3e} |")


if __name__ == "__main__":
    main()
