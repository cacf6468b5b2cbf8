#!/usr/bin/env python3
"""G2 MSM of 2^16 pairs over registered bases for several table geometries (tables x window bits): 17 x 15 wins (2^14 buckets:
the fold is one round), which is what snarkvm_amd/proofs.py registers."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snarkvm_amd import synthetic
from snarkvm_amd.msm import RegisteredBasesG2
n = 1 << 16
bases = synthetic.g2_points(n)
sc = synthetic.random_fr_integers(n, 1616)
for tables, bits in [(16, 0), (17, 15), (19, 14), (20, 13), (22, 12), (24, 11)]:
    rg = RegisteredBasesG2(bases, tables=tables, window_bits=bits)
    for _ in range(3): rg.msm(sc)
    t0 = time.perf_counter()
    for _ in range(10): rg.msm(sc)
    print(tables, bits, f"{(time.perf_counter()-t0)/10*1e3:.3f} ms", flush=True)
    rg.close()
