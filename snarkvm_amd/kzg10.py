"""`KZG10::{commit, commit_lagrange}` (algorithms/src/polycommit/kzg10/mod.rs:98-206) on the gfx950 backend.

The committer key's base vectors (`powers_of_beta_g`, `powers_of_beta_times_gamma_g`; data_structures.rs:151-181) are
registered ONCE in HBM in the kernels' native format - the reference's GPU path re-uploads 104 B/point on every call
(algorithms/cuda/cuda/snarkvm.cu:262-275).  A commitment is then a single fused device MSM:

    commit = msm(powers_of_beta_g[lz .. lz+len], to_bigint(coeffs[lz..]))            (mod.rs:110-120)
           + msm(powers_of_beta_times_gamma_g[.. h+1], to_bigint(blinding coeffs))     (mod.rs:146-150)

with `skip_leading_zeros_and_convert_to_bigints` (mod.rs:455-474) split into a host-side zero count and a
`Fr::to_bigint` fused into the MSM's scalar-read kernel (no separate pass, nothing leaves HBM).  The result is the
projective commitment; `KZGCommitment(commitment.into())` is one affine normalisation (`to_affine`).

Degree-bounded commitments (sonic_pc) use the same call with `powers` = the shifted powers slice.
"""
import ctypes

import numpy as np

from . import _lib
from .layout import G1_AFFINE, G1_PROJECTIVE


class PCError(ValueError):
    pass


def to_affine(projective):
    """`From<Projective> for Affine` on the device (affine.rs:331-353)."""
    projective = np.ascontiguousarray(projective, dtype=G1_PROJECTIVE).reshape(-1)
    out = np.zeros(projective.shape[0], dtype=G1_AFFINE)
    _lib.check(_lib.lib().snarkvm_hip_g1_to_affine(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(projective.ctypes.data),
                                                   ctypes.c_size_t(projective.shape[0])))
    return out


class Powers:
    """kzg10/data_structures.rs:151-181 `Powers`: the two base vectors of a committer key, resident in HBM."""

    def __init__(self, powers_of_beta_g, powers_of_beta_times_gamma_g, tables=16):
        """tables: precomputed 2^(256/tables * j) multiples kept next to every base (see RegisteredBases)."""
        self.powers_of_beta_g = np.ascontiguousarray(powers_of_beta_g, dtype=G1_AFFINE).reshape(-1)
        self.powers_of_beta_times_gamma_g = np.ascontiguousarray(powers_of_beta_times_gamma_g, dtype=G1_AFFINE).reshape(-1)
        self._gamma_offset = self.powers_of_beta_g.shape[0]
        both = np.concatenate([self.powers_of_beta_g, self.powers_of_beta_times_gamma_g])
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().snarkvm_hip_register_bases_tables(ctypes.byref(self._h), ctypes.c_void_p(both.ctypes.data),
                                                               ctypes.c_size_t(both.shape[0]), ctypes.c_size_t(G1_AFFINE.itemsize), ctypes.c_int(0),
                                                               ctypes.c_int(int(tables))))

    def size(self):  # data_structures.rs:163-165
        return self.powers_of_beta_g.shape[0]

    def close(self):
        if self._h:
            _lib.lib().snarkvm_hip_free_bases(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KZGRandomness:
    """data_structures.rs:313-376: the blinding polynomial (coefficient vector, Montgomery limbs)."""

    def __init__(self, blinding_coeffs=None):
        self.blinding_polynomial = np.zeros((0, 4), dtype=np.uint64) if blinding_coeffs is None else blinding_coeffs

    @classmethod
    def empty(cls):
        return cls()


class KZG10:
    @staticmethod
    def commit(powers, coeffs, hiding_bound=None, rng=None):
        """mod.rs:98-156.  `coeffs`: (d+1, 4) u64 Montgomery limbs of a dense polynomial (trailing zeros trimmed, like
        DensePolynomial).  `rng(k)` must return k uniformly random Fr elements as (k,4) Montgomery limbs.
        Returns (commitment as G1_PROJECTIVE record, KZGRandomness)."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        degree = max(coeffs.shape[0] - 1, 0)
        if degree + 1 > powers.size() and coeffs.shape[0] > 0:  # check_degree_is_too_large (mod.rs:407-415)
            raise PCError(f"TooManyCoefficients: {degree + 1} > {powers.size()}")
        # skip_leading_zeros_and_convert_to_bigints (mod.rs:455-467): count the leading zero coefficients
        nz = np.nonzero(coeffs.any(axis=1))[0]
        if nz.size == 0:
            lz, plain = 0, coeffs[:0]
        else:
            lz, plain = int(nz[0]), coeffs[int(nz[0]):]
        randomness = KZGRandomness.empty()
        if hiding_bound is not None:
            if rng is None:
                raise PCError("MissingRng")
            # KZGRandomness::rand(hiding_degree, false, rng) samples hiding_degree + 1 coefficients (data_structures.rs:344-350)
            randomness = KZGRandomness(np.ascontiguousarray(rng(hiding_bound + 1), dtype=np.uint64).reshape(-1, 4))
            deg = randomness.blinding_polynomial.shape[0] - 1
            if deg + 1 > powers.powers_of_beta_times_gamma_g.shape[0]:  # check_hiding_bound (mod.rs:417-427)
                raise PCError("HidingBoundToolarge")
        blind = randomness.blinding_polynomial
        scalars = np.concatenate([plain, blind]) if blind.shape[0] else plain
        out = np.zeros(1, dtype=G1_PROJECTIVE)
        _lib.check(_lib.lib().snarkvm_hip_msm_registered_ex(
            ctypes.c_void_p(out.ctypes.data), powers._h, ctypes.c_size_t(lz), ctypes.c_size_t(plain.shape[0]),
            ctypes.c_size_t(powers._gamma_offset), ctypes.c_size_t(blind.shape[0]),
            ctypes.c_void_p(np.ascontiguousarray(scalars).ctypes.data), ctypes.c_int(0), ctypes.c_int(1), ctypes.c_int(0)))
        return out, randomness

    @staticmethod
    def commit_lagrange(lagrange_basis, evaluations, hiding_bound=None, rng=None):
        """mod.rs:159-206: same MSM shape over `lagrange_basis_at_beta_g` (pass it as `Powers.powers_of_beta_g`);
        the evaluation vector is not trimmed and must fill the basis' power-of-two size."""
        evaluations = np.ascontiguousarray(evaluations, dtype=np.uint64).reshape(-1, 4)
        n = evaluations.shape[0]
        size = 1
        while size < n:
            size <<= 1
        if size != lagrange_basis.size():
            raise PCError("LagrangeBasisSizeIsIncorrect")
        randomness = KZGRandomness.empty()
        if hiding_bound is not None:
            if rng is None:
                raise PCError("MissingRng")
            randomness = KZGRandomness(np.ascontiguousarray(rng(hiding_bound + 1), dtype=np.uint64).reshape(-1, 4))
        blind = randomness.blinding_polynomial
        scalars = np.concatenate([evaluations, blind]) if blind.shape[0] else evaluations
        out = np.zeros(1, dtype=G1_PROJECTIVE)
        _lib.check(_lib.lib().snarkvm_hip_msm_registered_ex(
            ctypes.c_void_p(out.ctypes.data), lagrange_basis._h, ctypes.c_size_t(0), ctypes.c_size_t(n),
            ctypes.c_size_t(lagrange_basis._gamma_offset), ctypes.c_size_t(blind.shape[0]),
            ctypes.c_void_p(np.ascontiguousarray(scalars).ctypes.data), ctypes.c_int(0), ctypes.c_int(1), ctypes.c_int(0)))
        return out, randomness
