#!/usr/bin/env python3
"""Diagnostic for the lazy G2 path: tiny sizes, progress lines, meant to run under `timeout`."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import cpu as oracle
from snarkvm_amd import _lib, synthetic
from snarkvm_amd.msm import RegisteredBasesG2, msm_g2

def p(*a):
    print(*a, flush=True)

pts = synthetic.g2_points(4096, distinct=64)
for n in (1, 2, 3, 64, 1000, 4096):
    sc = synthetic.random_fr_integers(n, 100 + n)
    if n <= 3:
        sc[:, 1:] = 0
        sc[:, 0] &= np.uint64(0xFF)
    p("one-shot", n, "...")
    t0 = time.time()
    got = msm_g2(pts[:n], sc)
    p("   done in", round(time.time() - t0, 3), "s; checking")
    want = oracle.g2_msm(pts[:n], sc)
    p("   match:", oracle.g2_to_affine(got).tobytes() == oracle.g2_to_affine(want).tobytes())
for n in (64, 4096):
    sc = synthetic.random_fr_integers(n, 200 + n)
    p("register", n)
    rb = RegisteredBasesG2(pts[:n], tables=17, window_bits=15)
    p("registered msm", n, "...")
    t0 = time.time()
    got = rb.msm(sc)
    p("   done in", round(time.time() - t0, 3), "s")
    want = oracle.g2_msm(pts[:n], sc)
    p("   match:", oracle.g2_to_affine(got).tobytes() == oracle.g2_to_affine(want).tobytes())
    rb.close()
p("DIAG_DONE")
