#!/usr/bin/env python3
"""How much does the HIP runtime's pageable-copy path depend on whether it has seen the host pages before?  Times
`snarkvm_msm` (cached bases: only the scalars cross PCIe) with (a) the same scalar buffer every call, (b) a freshly
allocated copy per call (what a Rust caller's `convert_to_bigints` Vec is), (c) a fresh copy whose pages came from a reused
heap block (malloc arena reuse)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from snarkvm_amd import _lib, plugin, synthetic
from snarkvm_amd.layout import G1_AFFINE

L = _lib.lib(); torch.cuda.set_device(0)
for lg in [int(a) for a in sys.argv[1:]] or [20, 24]:
    n = 1 << lg
    buf = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(n)))
    bases = buf.cpu().numpy().view(G1_AFFINE); del buf
    sc = synthetic.random_fr_integers(n, 5)
    for _ in range(3): plugin.msm(bases, sc)          # registers the base range (second sighting)
    def t(f, reps=3):
        out = []
        for _ in range(reps):
            a = f(); t0 = time.perf_counter(); plugin.msm(bases, a); out.append((time.perf_counter() - t0) * 1e3)
        return " ".join(f"{x:.2f}" for x in out)
    print(f"2^{lg} same scalar buffer      :", t(lambda: sc), "ms", flush=True)
    print(f"2^{lg} fresh np.copy per call  :", t(lambda: sc.copy()), "ms", flush=True)
    pool = np.empty_like(sc)
    def reuse():
        pool[:] = sc; return pool
    print(f"2^{lg} rewritten reused buffer :", t(reuse), "ms", flush=True)
    print(f"2^{lg} fresh bases + same scalars (uncached path):", end=" ")
    for _ in range(2):
        b2 = bases.copy(); t0 = time.perf_counter(); plugin.msm(b2, sc); print(f"{(time.perf_counter()-t0)*1e3:.2f}", end=" ", flush=True)
    print("ms")
