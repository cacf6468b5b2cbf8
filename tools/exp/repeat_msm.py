"""diag: plain synchronous registered MSM (g1|g2), n = 2^lg, repeated; distinct affine results.  ROOT env: which checkout's snarkvm_amd to import."""
import ctypes, os, sys, collections
import numpy as np
ROOT = os.environ.get("ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from snarkvm_amd import _lib, synthetic
from snarkvm_amd.layout import G2_PROJECTIVE, G1_PROJECTIVE
sys.path.insert(0, '/root/repo')
import importlib.util
spec = importlib.util.spec_from_file_location("oracle_cpu", "/root/repo/oracle/cpu.py")
import torch
L = _lib.lib()
grp = sys.argv[1]; lg = int(sys.argv[2]); reps = int(sys.argv[3]); tables = int(sys.argv[4]); wb = int(sys.argv[5])
n = 1 << lg
sc = synthetic.random_fr_integers(n, 4)
d_sc = torch.from_numpy(sc.view(np.uint8).reshape(-1).copy()).cuda()
h = ctypes.c_void_p()
if grp == "g2":
    pts = synthetic.g2_points(n, distinct=int(os.environ.get("DISTINCT", "512")))
    _lib.check(L.snarkvm_hip_register_bases_g2(ctypes.byref(h), ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(n), ctypes.c_size_t(pts.dtype.itemsize), tables, wb))
    out = np.zeros(1, dtype=G2_PROJECTIVE)
    fn = L.snarkvm_hip_msm_g2_registered
else:
    from snarkvm_amd.layout import G1_AFFINE
    d = min(n, int(os.environ.get("DISTINCT", "512")))
    buf = torch.empty(d * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(d)))
    pts = np.tile(buf.cpu().numpy().view(G1_AFFINE), (n + d - 1) // d)[:n].copy()
    _lib.check(L.snarkvm_hip_register_bases_windowed(ctypes.byref(h), ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(n), ctypes.c_size_t(pts.dtype.itemsize), 0, tables, wb))
    out = np.zeros(1, dtype=G1_PROJECTIVE)
    fn = L.snarkvm_hip_msm_registered
from oracle import cpu as oracle
cnt = collections.Counter()
for i in range(reps):
    out[:] = np.zeros(1, dtype=out.dtype)
    _lib.check(fn(ctypes.c_void_p(out.ctypes.data), h, ctypes.c_size_t(0), ctypes.c_size_t(n), ctypes.c_void_p(d_sc.data_ptr()), 1, 0))
    cnt[(oracle.g2_to_affine(out) if grp == "g2" else oracle.g1_to_affine(out)).tobytes()] += 1
print(ROOT, os.environ.get("SNARKVM_HIP_TUNING", "default"), grp, "lg", lg, "tables", tables, "wb", wb, "distinct", len(cnt), "counts", sorted(cnt.values(), reverse=True)[:6], flush=True)
