"""What tuning fuse_reduce trades: a fused group of K proof-sized G1 instances whose scalars are ALL EQUAL (every entry of a digit row in one bucket) against uniform ones.
    SNARKVM_HIP_TUNING=fuse_reduce=1 python tools/exp/skew_group.py ; SNARKVM_HIP_TUNING=fuse_reduce=-1 python tools/exp/skew_group.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cpu as oracle
from snarkvm_amd import msm, synthetic
from tests import util

n = 1 << 17
bases = oracle.g1_gen_bases(util.g1_generator_affine(), 1, n)
rb = msm.RegisteredBases(bases, tables=17, window_bits=15)
uni = synthetic.random_fr_integers(n, 31)
one = np.zeros((n, 4), dtype=np.uint64); one[:, 0] = 1
same = np.tile(uni[:1], (n, 1))
half = uni.copy(); half[::2] = one[::2]
for name, sc in (("uniform", uni), ("all ones", one), ("all equal to one random value", same), ("every other scalar = 1", half)):
    for K in (1, 4):
        want = oracle.g1_to_affine(oracle.g1_msm(bases, sc)).tobytes() if K == 1 else None
        for _ in range(3):
            res = rb.msm_batch([sc] * K)
        if want is not None:
            assert oracle.g1_to_affine(res[0:1]).tobytes() == want, name
        t = time.perf_counter()
        for _ in range(10):
            rb.msm_batch([sc] * K)
        print(os.environ.get("SNARKVM_HIP_TUNING", "default"), f"| {name} | group of {K} x 2^17 | {(time.perf_counter() - t) * 100:.3f} ms per call", flush=True)
rb.close()
