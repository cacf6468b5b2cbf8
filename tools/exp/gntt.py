import sys, time, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np
from snarkvm_amd import _lib, group
from snarkvm_amd.layout import G1_AFFINE, G1_PROJECTIVE
from snarkvm_amd.devmem import HipMem
L = _lib.lib()
nmax = 1 << 13
d = HipMem(nmax * 104)
_lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(d.ptr), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
aff = d.download().view(G1_AFFINE)
for lg in (5, 6, 7, 8, 9, 10, 11, 12, 13):
    n = 1 << lg
    proj = np.zeros(n, dtype=G1_PROJECTIVE)
    proj["x"], proj["y"] = aff[:n]["x"], aff[:n]["y"]
    proj["z"] = np.array([202099033278250856, 5854854902718660529, 11492539364873682930, 8885205928937022213, 5545221690922665192, 39800542322357402], dtype=np.uint64)
    t0 = time.perf_counter()
    group.group_ntt(proj, inverse=True)
    print(lg, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
