#!/usr/bin/env python3
"""HIP-event phases of the reference's own `snarkvm_msm` symbol (host buffers, SNARKVM_HIP_BASE_CACHE=0: upload, conversion
and a table-less MSM on every call) at the small sizes a prover issues.  Prints a markdown table."""
import os, sys, time
os.environ.setdefault("SNARKVM_HIP_BASE_CACHE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np, torch
from snarkvm_amd import _lib, plugin, synthetic
from snarkvm_amd.layout import G1_AFFINE

L = _lib.lib(); torch.cuda.set_device(0)
sizes = [int(a) for a in sys.argv[1:]] or [12, 14, 16, 18]
nmax = 1 << max(sizes)
buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
_lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
bases = buf.cpu().numpy().view(G1_AFFINE); del buf
sc = synthetic.random_fr_integers(nmax, 5)
for lg in sizes:
    n = 1 << lg
    for _ in range(3): plugin.msm(bases[:n], sc[:n])
    t0 = time.perf_counter()
    for _ in range(20): plugin.msm(bases[:n], sc[:n])
    wall = (time.perf_counter() - t0) / 20 * 1e3
    L.snarkvm_hip_set_profiling(1)
    ph, order = {}, []
    for _ in range(5):
        plugin.msm(bases[:n], sc[:n])
        for i in range(L.snarkvm_hip_get_phase_count()):
            name = L.snarkvm_hip_get_phase_name(i).decode()
            if name not in ph: order.append(name)
            ph[name] = ph.get(name, 0.0) + L.snarkvm_hip_get_phase_ms(i) / 5
    L.snarkvm_hip_set_profiling(0)
    print(f"### snarkvm_msm 2^{lg}, table-less: {wall:.3f} ms per call; phases sum {sum(ph.values()):.3f} ms")
    print("| " + " | ".join(order) + " |"); print("|" + "---|" * len(order)); print("| " + " | ".join(f"{ph[k]:.3f}" for k in order) + " |\n", flush=True)
