import hashlib, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import cpu as oracle
from snarkvm_amd import _lib, synthetic
from snarkvm_amd.msm import RegisteredBasesG2
n = 1 << 16
bases = synthetic.g2_points(n)
sc = synthetic.random_fr_integers(n, 4016)
rb = RegisteredBasesG2(bases, tables=17, window_bits=15)
hs, affs = [], []
for i in range(8):
    r = rb.msm(sc)
    hs.append(hashlib.sha256(r.tobytes()).hexdigest()[:8])
    affs.append(hashlib.sha256(oracle.g2_to_affine(r).tobytes()).hexdigest()[:8])
print(os.environ.get("SNARKVM_HIP_TUNING", ""), "raw:", hs, "affine:", affs)
rb.close()
