"""diag: registered G2 MSM over few distinct points vs the oracle, several geometries; and the host-buffer G2 MSM over copies of one point."""
import ctypes, os, sys, collections
import numpy as np
ROOT = os.environ.get("ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from snarkvm_amd import _lib, synthetic, msm
from snarkvm_amd.layout import G2_PROJECTIVE, G2_AFFINE
sys.path.insert(0, '/root/repo')
from oracle import cpu as oracle
from oracle import pyref
L = _lib.lib()
n = 1024
for distinct in (1, 2, 16, 512):
    pts = synthetic.g2_points(n, distinct=distinct)
    sc = synthetic.random_fr_integers(n, 77 + distinct)
    want = oracle.g2_to_affine(oracle.g2_msm(pts.view(oracle.G2_AFFINE), sc, oracle.MSM_STANDARD)).tobytes()
    for tables, wb in ((1, 0), (2, 0), (16, 0), (17, 15)):
        rg = msm.RegisteredBasesG2(pts, tables=tables, window_bits=wb) if wb else msm.RegisteredBasesG2(pts, tables=tables)
        ok = sum(oracle.g2_to_affine(rg.msm(sc)).tobytes() == want for _ in range(10))
        rg.close()
        print(os.path.basename(ROOT), f"registered n={n} distinct={distinct} tables={tables} wb={wb}: {ok}/10 right", flush=True)
    ok = sum(oracle.g2_to_affine(msm.msm_g2(pts, sc)).tobytes() == want for _ in range(5))
    print(os.path.basename(ROOT), f"host-buffer n={n} distinct={distinct}: {ok}/5 right", flush=True)
for m2 in (1 << 12, (1 << 16) + 5):
    pts = synthetic.g2_points(m2, distinct=1)
    sc = synthetic.random_fr_integers(m2, 99)
    ksum = 0
    for limb in range(4):
        ksum += int(np.sum(sc[:, limb].astype(object))) << (64 * limb)
    want = oracle.g2_to_affine(oracle.g2_msm(pts[:1].view(oracle.G2_AFFINE), np.array([[(ksum % pyref.R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64))).tobytes()
    ok = sum(oracle.g2_to_affine(msm.msm_g2(pts, sc)).tobytes() == want for _ in range(5))
    print(os.path.basename(ROOT), f"host-buffer copies of one point m={m2}: {ok}/5 right", flush=True)
