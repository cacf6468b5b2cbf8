"""`KZG10::{commit, commit_lagrange, open, open_lagrange}` (algorithms/src/polycommit/kzg10/mod.rs:98-322) on the gfx950 backend.

The committer key's base vectors (`powers_of_beta_g`, `powers_of_beta_times_gamma_g`; data_structures.rs:151-181) are
registered ONCE in HBM in the kernels' native format - the reference's GPU path re-uploads 104 B/point on every call
(algorithms/cuda/cuda/snarkvm.cu:262-275).  A commitment is then a single fused device MSM:

    commit = msm(powers_of_beta_g[lz .. lz+len], to_bigint(coeffs[lz..]))            (mod.rs:110-120)
           + msm(powers_of_beta_times_gamma_g[.. h+1], to_bigint(blinding coeffs))     (mod.rs:146-150)

with `skip_leading_zeros_and_convert_to_bigints` (mod.rs:455-474) split into a host-side zero count and a
`Fr::to_bigint` fused into the MSM's scalar-read kernel (no separate pass, nothing leaves HBM).  The result is the
projective commitment; `KZGCommitment(commitment.into())` is one affine normalisation (`to_affine`).

Degree-bounded commitments (sonic_pc) use the same call with `powers` = the shifted powers slice.

An opening (mod.rs:213-322) is the quotient by (X - point) - a device suffix-Horner pass that also yields p(point)
(poly.divide_by_linear) - followed by the same fused MSM over the witness polynomial and, when hiding, the quotient of
the blinding polynomial over the gamma powers.
"""
import ctypes

import numpy as np

from . import _lib, poly
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

    @staticmethod
    def calculate_hiding_polynomial_degree(hiding_bound):  # data_structures.rs:340-343
        return hiding_bound + 1

    @classmethod
    def rand(cls, hiding_bound, rng):
        """data_structures.rs:351-356 -> DensePolynomial::rand(hiding_bound + 1) (fft/polynomial/dense.rs:120-127): a polynomial
        of DEGREE hiding_bound + 1, i.e. hiding_bound + 2 uniformly random coefficients, the leading one re-sampled while it
        is zero.  `rng(k)` returns k random Fr elements as (k, 4) Montgomery limbs."""
        d = cls.calculate_hiding_polynomial_degree(hiding_bound)
        c = np.array(rng(d + 1), dtype=np.uint64, copy=True).reshape(-1, 4)
        if c.shape[0] != d + 1:
            raise PCError(f"rng returned {c.shape[0]} elements, {d + 1} requested")
        while not c[d].any():  # "In the extremely unlikely event, sample again."
            c[d] = np.asarray(rng(1), dtype=np.uint64).reshape(-1, 4)[0]
        return cls(np.ascontiguousarray(c))

    def degree(self):
        return max(self.blinding_polynomial.shape[0] - 1, 0)

    def is_hiding(self):  # data_structures.rs:335-337: the blinding polynomial is non-zero
        return bool(np.asarray(self.blinding_polynomial).any())


def check_hiding_bound(hiding_poly_degree, num_powers):
    """kzg10/mod.rs:417-427: committing to a hiding polynomial of degree d needs d + 1 powers of beta * gamma * G."""
    if hiding_poly_degree == 0:
        raise PCError("HidingBoundIsZero")
    if hiding_poly_degree >= num_powers:
        raise PCError(f"HidingBoundToolarge: hiding_poly_degree {hiding_poly_degree}, num_powers {num_powers}")


class KZGProof:
    """data_structures.rs:395-430: w = commitment to the witness polynomial (G1Affine record),
    random_v = evaluation of the blinding polynomial at the point ((1, 4) Montgomery limbs) or None."""

    def __init__(self, w, random_v=None):
        self.w = w
        self.random_v = random_v

    def is_hiding(self):
        return self.random_v is not None


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
            randomness = KZGRandomness.rand(hiding_bound, rng)  # degree hiding_bound + 1 (data_structures.rs:351-356)
            check_hiding_bound(randomness.degree(), powers.powers_of_beta_times_gamma_g.shape[0])  # mod.rs:138-141
        blind = randomness.blinding_polynomial
        scalars = np.concatenate([plain, blind]) if blind.shape[0] else plain
        out = np.zeros(1, dtype=G1_PROJECTIVE)
        _lib.check(_lib.lib().snarkvm_hip_msm_registered_ex(
            ctypes.c_void_p(out.ctypes.data), powers._h, ctypes.c_size_t(lz), ctypes.c_size_t(plain.shape[0]),
            ctypes.c_size_t(powers._gamma_offset), ctypes.c_size_t(blind.shape[0]),
            ctypes.c_void_p(np.ascontiguousarray(scalars).ctypes.data), ctypes.c_int(0), ctypes.c_int(1), ctypes.c_int(0)))
        return out, randomness

    @staticmethod
    def commit_device(powers, coeffs, hiding_bound=None, rng=None):
        """`commit` for a coefficient vector that already lives in HBM (e.g. the output of a device iNTT): `coeffs` is a
        CUDA torch tensor viewing n Fr elements (any dtype, 32 * n bytes).  Nothing returns to the host but the 144-byte
        result: `to_bigint` is fused into the MSM's scalar read, leading zeros need no special case on the device (a zero
        digit touches no bucket), and the blinding coefficients are appended in HBM.  Same result as `commit`."""
        import torch

        nbytes = coeffs.numel() * coeffs.element_size()
        if nbytes % 32 or not coeffs.is_cuda or not coeffs.is_contiguous():
            raise PCError("commit_device: need a contiguous CUDA tensor of 32-byte Fr elements")
        n = nbytes // 32
        if n > powers.size():
            raise PCError(f"TooManyCoefficients: {n} > {powers.size()}")
        randomness = KZGRandomness.empty()
        flat = coeffs.view(torch.uint8).reshape(-1)
        k = 0
        if hiding_bound is not None:
            if rng is None:
                raise PCError("MissingRng")
            randomness = KZGRandomness.rand(hiding_bound, rng)
            k = randomness.blinding_polynomial.shape[0]
            check_hiding_bound(randomness.degree(), powers.powers_of_beta_times_gamma_g.shape[0])
            blind = torch.from_numpy(randomness.blinding_polynomial.view(np.uint8).reshape(-1).copy()).to(flat.device)
            flat = torch.cat([flat, blind])
        torch.cuda.current_stream().synchronize()  # the library runs on its own stream
        out = np.zeros(1, dtype=G1_PROJECTIVE)
        _lib.check(_lib.lib().snarkvm_hip_msm_registered_ex(
            ctypes.c_void_p(out.ctypes.data), powers._h, ctypes.c_size_t(0), ctypes.c_size_t(n),
            ctypes.c_size_t(powers._gamma_offset), ctypes.c_size_t(k), ctypes.c_void_p(flat.data_ptr()), ctypes.c_int(1), ctypes.c_int(1),
            ctypes.c_int(0)))
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
            randomness = KZGRandomness.rand(hiding_bound, rng)  # mod.rs:186-192
            check_hiding_bound(randomness.degree(), lagrange_basis.powers_of_beta_times_gamma_g.shape[0])
        blind = randomness.blinding_polynomial
        scalars = np.concatenate([evaluations, blind]) if blind.shape[0] else evaluations
        out = np.zeros(1, dtype=G1_PROJECTIVE)
        _lib.check(_lib.lib().snarkvm_hip_msm_registered_ex(
            ctypes.c_void_p(out.ctypes.data), lagrange_basis._h, ctypes.c_size_t(0), ctypes.c_size_t(n),
            ctypes.c_size_t(lagrange_basis._gamma_offset), ctypes.c_size_t(blind.shape[0]),
            ctypes.c_void_p(np.ascontiguousarray(scalars).ctypes.data), ctypes.c_int(0), ctypes.c_int(1), ctypes.c_int(0)))
        return out, randomness

    @staticmethod
    def compute_witness_polynomial(coeffs, point, randomness):
        """mod.rs:213-236: (p / (X - point), blinding_p / (X - point) if hiding else None); remainders are dropped."""
        witness, _ = poly.divide_by_linear(coeffs, point)
        random_witness = None
        if randomness.is_hiding():
            random_witness, _ = poly.divide_by_linear(randomness.blinding_polynomial, point)
        return witness, random_witness

    @staticmethod
    def open_with_witness_polynomial(powers, point, randomness, witness_polynomial, hiding_witness_polynomial=None):
        """mod.rs:238-270.  One fused MSM: witness coefficients over powers_of_beta_g[lz..], the hiding witness over
        powers_of_beta_times_gamma_g; `to_affine` gives KZGProof.w."""
        witness = poly.trim(witness_polynomial)
        degree = max(witness.shape[0] - 1, 0)
        if degree + 1 > powers.size():  # check_degree_is_too_large (mod.rs:407-415)
            raise PCError(f"TooManyCoefficients: {degree + 1} > {powers.size()}")
        nz = np.nonzero(witness.any(axis=1))[0]
        lz, plain = (0, witness[:0]) if nz.size == 0 else (int(nz[0]), witness[int(nz[0]):])
        random_v = None
        blind = np.zeros((0, 4), dtype=np.uint64)
        if hiding_witness_polynomial is not None:
            random_v = poly.evaluate(randomness.blinding_polynomial, point)  # blinding_p.evaluate(point), mod.rs:254
            blind = np.ascontiguousarray(hiding_witness_polynomial, dtype=np.uint64).reshape(-1, 4)  # untrimmed: convert_to_bigints(&coeffs)
            if blind.shape[0] > powers.powers_of_beta_times_gamma_g.shape[0]:
                raise PCError("HidingBoundToolarge")
        scalars = np.concatenate([plain, blind]) if blind.shape[0] else plain
        out = np.zeros(1, dtype=G1_PROJECTIVE)
        _lib.check(_lib.lib().snarkvm_hip_msm_registered_ex(
            ctypes.c_void_p(out.ctypes.data), powers._h, ctypes.c_size_t(lz), ctypes.c_size_t(plain.shape[0]),
            ctypes.c_size_t(powers._gamma_offset), ctypes.c_size_t(blind.shape[0]),
            ctypes.c_void_p(np.ascontiguousarray(scalars).ctypes.data), ctypes.c_int(0), ctypes.c_int(1), ctypes.c_int(0)))
        return KZGProof(to_affine(out)[0], random_v)

    @staticmethod
    def open(powers, coeffs, point, rand):
        """mod.rs:304-322."""
        coeffs = poly.trim(coeffs)
        degree = max(coeffs.shape[0] - 1, 0)
        if degree + 1 > powers.size():
            raise PCError(f"TooManyCoefficients: {degree + 1} > {powers.size()}")
        witness, hiding_witness = KZG10.compute_witness_polynomial(coeffs, point, rand)
        return KZG10.open_with_witness_polynomial(powers, point, rand, witness, hiding_witness)

    @staticmethod
    def open_lagrange(lagrange_basis, domain_elements, evaluations, point, evaluation_at_point):
        """mod.rs:274-301: witness evaluations (e_i - v) / (omega_i - point) by one batch inversion, then commit_lagrange.
        Raises when the point lies in the domain (a zero divisor evaluation)."""
        evaluations = np.ascontiguousarray(evaluations, dtype=np.uint64).reshape(-1, 4)
        domain_elements = np.ascontiguousarray(domain_elements, dtype=np.uint64).reshape(-1, 4)
        if max(evaluations.shape[0] - 1, 0) + 1 > lagrange_basis.size():
            raise PCError("TooManyCoefficients")
        divisor_evals = poly.vec_op("sub_scalar", domain_elements, scalar=point)
        if not divisor_evals.any(axis=1).all():
            raise PCError("Point cannot be in the domain")
        size = 1
        while size < evaluations.shape[0]:
            size <<= 1
        if size != lagrange_basis.size():
            raise PCError("`evaluations.len()` must equal `domain.size()`")
        if divisor_evals.shape[0] != evaluations.shape[0]:
            raise PCError("length mismatch")
        one = np.array([[0x7d1c7ffffffffff3, 0x7257f50f6ffffff2, 0x16d81575512c0fee, 0x0d4bda322bbb9a9d]], dtype=np.uint64)  # Fr R (fr.rs:158-163)
        divisor_evals = poly.batch_inversion_and_mul(divisor_evals, one)
        numer = poly.vec_op("sub_scalar", evaluations, scalar=evaluation_at_point)
        witness_evals = poly.vec_op("mul", divisor_evals, numer)
        comm, _ = KZG10.commit_lagrange(lagrange_basis, witness_evals, None, None)
        return KZGProof(to_affine(comm)[0], None)
