"""diag: a batch of K registered G2 MSMs (fused group), repeated; distinct affine results per instance."""
import ctypes, os, sys, collections
import numpy as np
ROOT = os.environ.get("ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from snarkvm_amd import _lib, synthetic
from snarkvm_amd.layout import G2_PROJECTIVE
sys.path.insert(0, '/root/repo')
import torch
from oracle import cpu as oracle
L = _lib.lib()
lg = int(sys.argv[1]); reps = int(sys.argv[2]); tables = int(sys.argv[3]); wb = int(sys.argv[4]); K = int(sys.argv[5])
n = 1 << lg
pts = synthetic.g2_points(n)
h = ctypes.c_void_p()
_lib.check(L.snarkvm_hip_register_bases_g2(ctypes.byref(h), ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(n), ctypes.c_size_t(pts.dtype.itemsize), tables, wb))
d = [torch.from_numpy(synthetic.random_fr_integers(n, 4 + k).view(np.uint8).reshape(-1).copy()).cuda() for k in range(K)]
out = np.zeros(K, dtype=G2_PROJECTIVE)
offs = (ctypes.c_size_t * K)(*([0] * K)); ns = (ctypes.c_size_t * K)(*([n] * K)); ptrs = (ctypes.c_void_p * K)(*[x.data_ptr() for x in d])
cnt = [collections.Counter() for _ in range(K)]
for i in range(reps):
    out[:] = np.zeros(K, dtype=G2_PROJECTIVE)
    _lib.check(L.snarkvm_hip_msm_g2_registered_batch(ctypes.c_void_p(out.ctypes.data), h, ctypes.c_size_t(K), offs, ns, ptrs, 1, 0))
    a = oracle.g2_to_affine(out)
    for k in range(K):
        cnt[k][a[k:k + 1].tobytes()] += 1
print(ROOT, os.environ.get("SNARKVM_HIP_TUNING", "default"), "batch", K, "lg", lg, tables, wb, "distinct per instance", [len(c) for c in cnt], flush=True)
