#!/usr/bin/env python3
"""Diagnostic: one G2 one-shot MSM of 64 pairs under SNARKVM_HIP_TRACE=2 (meant to run under `timeout`)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from snarkvm_amd import _lib, synthetic
from snarkvm_amd.msm import msm_g2
pts = synthetic.g2_points(64, distinct=64)
sc = synthetic.random_fr_integers(64, 164)
print("calling", flush=True)
msm_g2(pts, sc)
print("DONE", flush=True)
