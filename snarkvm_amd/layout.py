"""Host-side data layouts at the drop-in boundary (numpy views of the Rust in-memory types).

Sizes per SURVEY.md Appendix B: Fr / BigInteger256 32 B, Fq 48 B, G1Affine 104 B (x, y, infinity + 7 pad),
G1Projective 144 B (Jacobian X, Y, Z), G2Affine 200 B, G2Projective 288 B.  All field elements are
little-endian u64 limbs; coordinates are Montgomery residues, MSM scalars are canonical integers < r
(reference: curves/src/templates/short_weierstrass_jacobian/affine.rs:42-46, projective.rs:37-41,
algorithms/src/polycommit/kzg10/mod.rs:469-474).
"""
import numpy as np

FR_LIMBS = 4
FQ_LIMBS = 6
G1_AFFINE = np.dtype([("x", "<u8", 6), ("y", "<u8", 6), ("infinity", "u1"), ("pad", "u1", 7)])
G1_PROJECTIVE = np.dtype([("x", "<u8", 6), ("y", "<u8", 6), ("z", "<u8", 6)])
G2_AFFINE = np.dtype([("x", "<u8", 12), ("y", "<u8", 12), ("infinity", "u1"), ("pad", "u1", 7)])
G2_PROJECTIVE = np.dtype([("x", "<u8", 12), ("y", "<u8", 12), ("z", "<u8", 12)])
assert G1_AFFINE.itemsize == 104 and G1_PROJECTIVE.itemsize == 144
assert G2_AFFINE.itemsize == 200 and G2_PROJECTIVE.itemsize == 288


class NTTInputOutputOrder:  # algorithms/cuda/src/lib.rs:22-28
    NN, NR, RN, RR = 0, 1, 2, 3


class NTTDirection:  # algorithms/cuda/src/lib.rs:30-34
    Forward, Inverse = 0, 1


class NTTType:  # algorithms/cuda/src/lib.rs:36-40
    Standard, Coset = 0, 1
