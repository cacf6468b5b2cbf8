import hashlib, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import ctypes
from snarkvm_amd import _lib, synthetic
from snarkvm_amd.msm import RegisteredBases
from snarkvm_amd.devmem import HipMem
from snarkvm_amd.layout import G1_AFFINE
n = 1 << 16
L = _lib.lib()
d = HipMem(n * 104)
_lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(d.ptr), ctypes.c_uint64(1), ctypes.c_size_t(n)))
rb = RegisteredBases(device_ptr=d.ptr, npoints=n, tables=17, window_bits=15)
sc = synthetic.random_fr_integers(n, 4016)
hs = []
for i in range(8):
    r = rb.msm(sc)
    hs.append(hashlib.sha256(r.tobytes()).hexdigest()[:8])
print("G1", os.environ.get("SNARKVM_HIP_TUNING", ""), "raw:", hs)
rb.close()
