"""`SonicKZG10::{commit, combine_for_open, batch_open}` (algorithms/src/polycommit/sonic_pc/mod.rs:177-342) on the gfx950 backend.

The caller side of the MSM hot path.  The reference fans the commitments of one call out over a CPU pool, one
`KZG10::commit` / `KZG10::commit_lagrange` per labelled polynomial (mod.rs:186-245), each of which re-uploads its slice of the
SRS to the GPU.  Here the committer key's base vectors - plain powers, gamma powers, shifted powers (degree bounds), the gamma
powers of every enforced bound and the Lagrange bases - live in ONE registered device vector, and a call is ONE batched device
launch (`snarkvm_hip_msm_registered_batch_ex`): instance k = (plaintext base range, hiding base range, coefficients).  The
device fuses `Fr::to_bigint` into its scalar read (kzg10/mod.rs:455-474) and runs the instances concurrently on several
streams (and devices); a degree-bounded polynomial is the same MSM starting at the shifted-powers offset
`max_bound - degree_bound` (data_structures.rs:310-331).

The Fiat-Shamir sponge is out of scope: `batch_open` takes the challenges it would squeeze from any object
with `squeeze_short_nonnative_field_element()`.
"""
import ctypes

import numpy as np

from . import _lib, kzg10, poly
from .kzg10 import KZG10, KZGRandomness, PCError
from .layout import G1_AFFINE, G1_PROJECTIVE

FR_ONE = np.array([0x7D1C7FFFFFFFFFF3, 0x7257F50F6FFFFFF2, 0x16D81575512C0FEE, 0x0D4BDA322BBB9A9D], dtype=np.uint64)  # Fr R (fr.rs:158-163)


def _pts(a):
    return np.ascontiguousarray(a, dtype=G1_AFFINE).reshape(-1)


class LabeledPolynomial:
    """polycommit/sonic_pc/polynomial.rs: label, dense coefficient vector (Montgomery limbs, trimmed), optional degree bound
    and hiding bound."""

    def __init__(self, label, coeffs, degree_bound=None, hiding_bound=None):
        self.label = label
        self.coeffs = poly.trim(coeffs)
        self.degree_bound = degree_bound
        self.hiding_bound = hiding_bound

    def degree(self):  # DensePolynomial::degree: 0 for the zero polynomial
        return max(self.coeffs.shape[0] - 1, 0)

    def is_hiding(self):
        return self.hiding_bound is not None


class LabeledEvaluations:
    """`PolynomialWithBasis::Lagrange`: evaluations over a power-of-two domain, committed in the Lagrange basis."""

    def __init__(self, label, evaluations, hiding_bound=None):
        self.label = label
        self.evaluations = np.ascontiguousarray(evaluations, dtype=np.uint64).reshape(-1, 4)
        self.degree_bound = None
        self.hiding_bound = hiding_bound

    def degree(self):
        return max(self.evaluations.shape[0] - 1, 0)


class LabeledCommitment:
    def __init__(self, label, commitment, degree_bound):
        self.label = label
        self.commitment = commitment  # G1_AFFINE record (KZGCommitment)
        self.degree_bound = degree_bound


class CommitterUnionKey:
    """sonic_pc/data_structures.rs:274-343, with every base vector registered once in HBM (one handle, recorded offsets)."""

    def __init__(self, powers_of_beta_g, powers_of_beta_times_gamma_g, shifted_powers_of_beta_g=None,
                 shifted_powers_of_beta_times_gamma_g=None, enforced_degree_bounds=None, lagrange_bases_at_beta_g=None, tables=16):
        self.powers_of_beta_g = _pts(powers_of_beta_g)
        self.powers_of_beta_times_gamma_g = _pts(powers_of_beta_times_gamma_g)
        self.shifted_powers_of_beta_g = None if shifted_powers_of_beta_g is None else _pts(shifted_powers_of_beta_g)
        self.shifted_powers_of_beta_times_gamma_g = None if shifted_powers_of_beta_times_gamma_g is None else {
            int(b): _pts(v) for b, v in shifted_powers_of_beta_times_gamma_g.items()}
        self.enforced_degree_bounds = None if enforced_degree_bounds is None else sorted(int(b) for b in enforced_degree_bounds)
        self.lagrange_bases_at_beta_g = {} if lagrange_bases_at_beta_g is None else {int(s): _pts(v) for s, v in lagrange_bases_at_beta_g.items()}
        parts, self._off = [], {}
        cursor = 0

        def place(key, arr):
            nonlocal cursor
            self._off[key] = cursor
            parts.append(arr)
            cursor += arr.shape[0]

        place("powers", self.powers_of_beta_g)
        place("gamma", self.powers_of_beta_times_gamma_g)
        if self.shifted_powers_of_beta_g is not None:
            place("shifted", self.shifted_powers_of_beta_g)
            for b, v in sorted((self.shifted_powers_of_beta_times_gamma_g or {}).items()):
                place(("shifted_gamma", b), v)
        for s, v in sorted(self.lagrange_bases_at_beta_g.items()):
            place(("lagrange", s), v)
        allpts = np.concatenate(parts)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().snarkvm_hip_register_bases_tables(ctypes.byref(self._h), ctypes.c_void_p(allpts.ctypes.data), ctypes.c_size_t(allpts.shape[0]),
                                                               ctypes.c_size_t(G1_AFFINE.itemsize), ctypes.c_int(0), ctypes.c_int(int(tables))))

    # ---- the views the reference hands to KZG10 (host arrays + their place in the registered vector)
    def powers(self):  # data_structures.rs:302-307
        return _PowersView(self, self._off["powers"], self.powers_of_beta_g.shape[0], self._off["gamma"], self.powers_of_beta_times_gamma_g.shape[0])

    def shifted_powers(self, degree_bound=None):  # data_structures.rs:310-331
        if self.shifted_powers_of_beta_g is None or self.shifted_powers_of_beta_times_gamma_g is None:
            return None
        max_bound = self.enforced_degree_bounds[-1]
        if degree_bound is not None:
            assert degree_bound in self.enforced_degree_bounds
            bound, start = degree_bound, max_bound - degree_bound
        else:
            bound, start = max_bound, 0
        gam = self.shifted_powers_of_beta_times_gamma_g[bound]
        return _PowersView(self, self._off["shifted"] + start, self.shifted_powers_of_beta_g.shape[0] - start, self._off[("shifted_gamma", bound)], gam.shape[0])

    def lagrange_basis(self, size):  # data_structures.rs:335-341
        if size not in self.lagrange_bases_at_beta_g:
            return None
        return _PowersView(self, self._off[("lagrange", size)], size, self._off["gamma"], self.powers_of_beta_times_gamma_g.shape[0])

    def close(self):
        if self._h:
            _lib.lib().snarkvm_hip_free_bases(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PowersView:
    """kzg10::Powers / kzg10::LagrangeBasis as (offset, length) windows of the committer key's registered vector; quacks like
    snarkvm_amd.kzg10.Powers for KZG10.open."""

    def __init__(self, ck, off, n, gamma_off, gamma_n):
        self._h = ck._h
        self.off, self.n, self._gamma_offset, self.gamma_n = off, n, gamma_off, gamma_n
        self.powers_of_beta_times_gamma_g = np.empty(gamma_n, dtype=np.uint8)  # only its length is consulted (check_hiding_bound)

    def size(self):
        return self.n


def check_degrees_and_bounds(max_degree, enforced_degree_bounds, p):
    """kzg10/mod.rs:428-452."""
    bound = p.degree_bound
    if bound is None:
        return
    if enforced_degree_bounds is None or bound not in enforced_degree_bounds:
        raise PCError(f"UnsupportedDegreeBound({bound})")
    if bound < p.degree() or bound > max_degree:
        raise PCError(f"IncorrectDegreeBound: poly_degree {p.degree()}, degree_bound {bound}, max_degree {max_degree}, label {p.label}")


class SonicKZG10:
    @staticmethod
    def commit(max_degree, ck, polynomials, rng=None):
        """sonic_pc/mod.rs:177-257.  polynomials: LabeledPolynomial / LabeledEvaluations; rng(k) -> k random Fr elements as (k, 4)
        Montgomery limbs (required when a polynomial is hiding).  Returns ([LabeledCommitment], [KZGRandomness]), in input order.
        All commitments of the call are ONE batched device launch."""
        items = list(polynomials)
        off0, n0, off1, n1, scal, rands = [], [], [], [], [], []
        for p in items:
            check_degrees_and_bounds(max_degree, ck.enforced_degree_bounds, p)
            if isinstance(p, LabeledEvaluations):
                n = p.evaluations.shape[0]
                size = 1
                while size < n:
                    size <<= 1
                view = ck.lagrange_basis(size)
                if view is None:
                    raise PCError(f"UnsupportedLagrangeBasisSize({size})")
                if size != n:  # KZG10::commit_lagrange: evaluations.len().next_power_of_two() must be the basis size
                    raise PCError("LagrangeBasisSizeIsIncorrect")
                lz, plain = 0, p.evaluations
            else:
                view = ck.shifted_powers(p.degree_bound) if p.degree_bound is not None else ck.powers()
                if view is None:
                    raise PCError(f"UnsupportedDegreeBound({p.degree_bound})")
                if p.coeffs.shape[0] and p.degree() + 1 > view.size():  # check_degree_is_too_large (kzg10/mod.rs:407-415)
                    raise PCError(f"TooManyCoefficients: {p.degree() + 1} > {view.size()}")
                nz = np.nonzero(p.coeffs.any(axis=1))[0]  # skip_leading_zeros_and_convert_to_bigints (kzg10/mod.rs:455-467)
                lz, plain = (0, p.coeffs[:0]) if nz.size == 0 else (int(nz[0]), p.coeffs[int(nz[0]):])
            randomness = KZGRandomness.empty()
            if p.hiding_bound is not None:
                if rng is None:
                    raise PCError("MissingRng")
                randomness = KZGRandomness.rand(p.hiding_bound, rng)  # degree hiding_bound + 1 (data_structures.rs:351-356)
                kzg10.check_hiding_bound(randomness.degree(), view.gamma_n)  # kzg10/mod.rs:417-427
            blind = randomness.blinding_polynomial
            off0.append(view.off + lz)
            n0.append(plain.shape[0])
            off1.append(view._gamma_offset)
            n1.append(blind.shape[0])
            scal.append(np.ascontiguousarray(np.concatenate([plain, blind]) if blind.shape[0] else plain))
            rands.append(randomness)
        k = len(items)
        outs = np.zeros(k, dtype=G1_PROJECTIVE)
        if k:
            arr = lambda v: (ctypes.c_size_t * k)(*v)  # noqa: E731
            ptrs = (ctypes.c_void_p * k)(*[s.ctypes.data for s in scal])
            _lib.check(_lib.lib().snarkvm_hip_msm_registered_batch_ex(ctypes.c_void_p(outs.ctypes.data), ck._h, ctypes.c_size_t(k), arr(off0), arr(n0), arr(off1),
                                                                     arr(n1), ptrs, ctypes.c_int(0), ctypes.c_int(1), ctypes.c_int(0)))
        affine = kzg10.to_affine(outs) if k else np.zeros(0, dtype=G1_AFFINE)
        return [LabeledCommitment(p.label, affine[i], p.degree_bound) for i, p in enumerate(items)], rands

    @staticmethod
    def combine_polynomials(coeffs_polys_rands):
        """sonic_pc/mod.rs:548-564: sum_i coeff_i * poly_i and the same combination of the blinding polynomials (device AXPY)."""
        items = list(coeffs_polys_rands)

        def combine(vectors):
            length = max([v.shape[0] for _, v in vectors] + [0])
            acc = np.zeros((length, 4), dtype=np.uint64)
            for coeff, v in vectors:
                if v.shape[0] == 0:
                    continue
                padded = np.zeros((length, 4), dtype=np.uint64)
                padded[: v.shape[0]] = v
                if np.array_equal(np.asarray(coeff, dtype=np.uint64).reshape(4), FR_ONE):
                    acc = poly.vec_op("add", acc, padded)
                else:
                    acc = poly.vec_op("axpy", acc, padded, scalar=coeff)
            return poly.trim(acc)

        combined_poly = combine([(c, p) for c, p, _ in items])
        combined_rand = KZGRandomness(combine([(c, np.ascontiguousarray(r.blinding_polynomial, dtype=np.uint64).reshape(-1, 4)) for c, _, r in items]))
        return combined_poly, combined_rand

    @staticmethod
    def combine_for_open(max_degree, ck, labeled_polynomials, rands, fs_rng):
        """sonic_pc/mod.rs:259-282: one challenge per polynomial, squeezed in order."""
        polys, rands = list(labeled_polynomials), list(rands)
        if len(polys) != len(rands):
            raise PCError("length mismatch")
        to_combine = []
        for p, r in zip(polys, rands):
            check_degrees_and_bounds(max_degree, ck.enforced_degree_bounds, p)
            to_combine.append((fs_rng.squeeze_short_nonnative_field_element(), p.coeffs, r))
        return SonicKZG10.combine_polynomials(to_combine)

    @staticmethod
    def batch_open(max_degree, ck, labeled_polynomials, query_set, rands, fs_rng):
        """sonic_pc/mod.rs:286-342.  query_set: iterable of (label, (point_name, point)); returns the KZGProofs in the order of
        the sorted point names (the reference iterates a BTreeMap)."""
        polys, rands = list(labeled_polynomials), list(rands)
        if len(polys) != len(rands):
            raise PCError("length mismatch")
        poly_rand = {p.label: (p, r) for p, r in zip(polys, rands)}
        query_to_labels = {}
        for label, (point_name, point) in query_set:
            entry = query_to_labels.setdefault(point_name, (point, set()))
            entry[1].add(label)
        proofs = []
        for point_name in sorted(query_to_labels):
            point, labels = query_to_labels[point_name]
            qp, qr = [], []
            for label in sorted(labels):
                if label not in poly_rand:
                    raise PCError(f"MissingPolynomial {{ label: {label} }}")
                qp.append(poly_rand[label][0])
                qr.append(poly_rand[label][1])
            polynomial, rand = SonicKZG10.combine_for_open(max_degree, ck, qp, qr, fs_rng)
            fs_rng.squeeze_short_nonnative_field_element()  # `_randomizer` (mod.rs:331)
            proofs.append(KZG10.open(_OpenPowers(ck.powers()), polynomial, point, rand))
        return proofs


class _OpenPowers:
    """Adapter: KZG10.open (snarkvm_amd.kzg10) addresses `powers._h` from offset `lz` and the gamma powers at `_gamma_offset`."""

    def __init__(self, view):
        self._view = view
        self._h = view._h
        self._gamma_offset = view._gamma_offset
        self.powers_of_beta_times_gamma_g = view.powers_of_beta_times_gamma_g
        if view.off != 0:
            raise PCError("the plain powers must start the registered vector")

    def size(self):
        return self._view.size()
