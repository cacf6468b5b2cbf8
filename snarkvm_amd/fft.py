"""`EvaluationDomain<Fr>` and `PolyMultiplier` (algorithms/src/fft/domain.rs:83-221, fft/polynomial/multiplier.rs:70-134)
over the gfx950 NTT.  Vectors are (n, 4) u64 arrays of Montgomery limbs (the Rust `Vec<Fr>` memory image)."""
import numpy as np

from . import plugin
from .layout import NTTDirection, NTTInputOutputOrder, NTTType

FR_TWO_ADICITY = 47  # curves/src/bls12_377/fr.rs:109
FR_TWO_ADIC_ROOT_OF_UNITY = 8065159656716812877374967518403273466521432693661810619979959746626482506078  # fr.rs:110
FR_GENERATOR = 22  # fr.rs:126: the multiplicative generator, the coset shift of coset_fft (domain.rs:195-221)


def _fr(value):
    """A field element as its Rust memory image: one (1, 4) u64 row of Montgomery limbs (value * 2^256 mod r)."""
    from .synthetic import R_MOD

    m = (value % R_MOD) * (1 << 256) % R_MOD
    return np.array([[(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64)


class FFTPrecomputation:
    """`FFTPrecomputation` / `IFFTPrecomputation` (domain.rs:883-933): the n/2 twiddles of a domain.  The device keeps its own
    tables (ntt.hip.h), so the transforms ignore it; it exists so that code written against `*_with_pc` keeps its shape, and
    `roots` holds what the reference's struct holds."""

    def __init__(self, domain, roots, inverse=False):
        self.domain, self.roots, self.inverse = domain, roots, inverse

    def to_ifft_precomputation(self):  # domain.rs:899-911
        return self.domain.precompute_ifft()


class EvaluationDomain:
    def __init__(self, size, log_size_of_group):
        from .synthetic import R_MOD

        self.size = size
        self.log_size_of_group = log_size_of_group
        # domain.rs:83-98, 118-147: the constants the reference keeps beside the size (Montgomery memory images)
        gen = pow(FR_TWO_ADIC_ROOT_OF_UNITY, 1 << (FR_TWO_ADICITY - log_size_of_group), R_MOD) if log_size_of_group <= FR_TWO_ADICITY else 0
        self.size_as_field_element = _fr(size)
        self.size_inv = _fr(pow(size, -1, R_MOD))
        self.group_gen = _fr(gen)
        self.group_gen_inv = _fr(pow(gen, -1, R_MOD)) if gen else _fr(0)
        self.generator_inv = _fr(pow(FR_GENERATOR, -1, R_MOD))

    def roots_of_unity(self, root):
        """domain.rs:594-648: [root^i for i < size / 2] (size / 2 = 0 gives the empty vector), computed on the device."""
        from . import poly

        half = self.size // 2
        if half == 0:
            return np.zeros((0, 4), dtype=np.uint64)
        return poly.distribute_powers_and_mul_by_const(np.repeat(_fr(1), half, axis=0), root, _fr(1))

    def precompute_fft(self):  # domain.rs:360-365
        return FFTPrecomputation(self, self.roots_of_unity(self.group_gen))

    def precompute_ifft(self):  # domain.rs:367-372
        return FFTPrecomputation(self, self.roots_of_unity(self.group_gen_inv), inverse=True)

    def fft_with_pc(self, coeffs, pc=None):  # in_order_fft_with_pc (domain.rs:374-392): the table is the device's own
        return self.fft(coeffs)

    def ifft_with_pc(self, evals, pc=None):  # in_order_ifft_with_pc (domain.rs:403-420)
        return self.ifft(evals)

    @classmethod
    def new(cls, num_coeffs):
        """domain.rs:118-147: size = next_power_of_two(num_coeffs); None if the field has no such subgroup."""
        size = 1
        while size < num_coeffs:
            size <<= 1
        lg = size.bit_length() - 1
        if lg > FR_TWO_ADICITY:
            return None
        return cls(size, lg)

    def _resized(self, v):
        """domain.rs:171,187,218: `coeffs.resize(self.size(), zero)` - zero-pad or truncate."""
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((self.size, 4), dtype=np.uint64)
        k = min(self.size, v.shape[0])
        out[:k] = v[:k]
        return out

    def _run(self, v, direction, kind):
        x = self._resized(v)
        plugin.NTT(self.size, x, NTTInputOutputOrder.NN, direction, kind)
        return x

    def fft(self, coeffs):  # domain.rs:162-174
        return self._run(coeffs, NTTDirection.Forward, NTTType.Standard)

    def ifft(self, evals):  # domain.rs:177-192
        return self._run(evals, NTTDirection.Inverse, NTTType.Standard)

    def coset_fft(self, coeffs):  # domain.rs:195-207
        return self._run(coeffs, NTTDirection.Forward, NTTType.Coset)

    def coset_ifft(self, evals):  # domain.rs:210-221
        return self._run(evals, NTTDirection.Inverse, NTTType.Coset)


class PolyMultiplier:
    """multiplier.rs:27-134: collect coefficient-form polynomials and evaluation-form vectors, multiply."""

    def __init__(self):
        self.polynomials = []
        self.evaluations = []

    def add_polynomial(self, coeffs, label=""):
        self.polynomials.append((label, np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)))

    def add_evaluation(self, evals, label=""):
        self.evaluations.append((label, np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)))

    def multiply(self):
        """Returns the coefficient vector with trailing zeros trimmed (DensePolynomial::from_coefficients_vec,
        dense.rs:76-86), or None when an evaluation vector lives on a different domain (multiplier.rs:76-77)."""
        if not self.polynomials and not self.evaluations:
            return np.zeros((0, 4), dtype=np.uint64)  # the zero polynomial
        degree = sum(max(p.shape[0], 1) for _, p in self.polynomials)  # sum(deg + 1) for trimmed inputs
        domain = EvaluationDomain.new(degree)
        if domain is None:
            return None
        if any(e.shape[0] != domain.size for _, e in self.evaluations):
            return None
        res = plugin.polymul(domain.size, [p for _, p in self.polynomials], [e for _, e in self.evaluations])
        nz = np.nonzero(res.any(axis=1))[0]
        return res[: (nz[-1] + 1 if nz.size else 0)]


    def element_wise_arithmetic_4_over_domain(self, domain, labels, f):
        """multiplier.rs:136-178: the four labelled inputs on `domain` - polynomials through a forward transform (resized to the
        domain), evaluation vectors resized - combined element by element by `f(a, b, c, d)` and interpolated back; returns the
        trimmed coefficient vector.  `f` receives and returns (|domain|, 4) arrays (e.g. built from `snarkvm_amd.poly.vec_op`,
        which runs on the device).  The reference evaluates in bit-reversed order (out-of-order FFT, `derange` of the
        evaluation vectors, out-of-order iFFT); an element-wise `f` commutes with that permutation, so the natural-order
        transforms give the same polynomial.  Like the reference it insists on exactly four inputs (`assert_eq!(p.len(), 4)`)."""
        vecs = {}
        for label, pol in self.polynomials:
            vecs[label] = domain.fft(pol)  # `_resized`: zero-pad or truncate to the domain (multiplier.rs:155)
        for label, ev in self.evaluations:
            vecs[label] = domain._resized(ev)  # multiplier.rs:163
        if len(vecs) != 4:
            raise AssertionError("element_wise_arithmetic_4_over_domain needs exactly four distinctly labelled inputs")
        result = np.ascontiguousarray(f(*[vecs[l] for l in labels]), dtype=np.uint64).reshape(-1, 4)
        coeffs = domain.ifft(result)
        nz = np.nonzero(coeffs.any(axis=1))[0]
        return coeffs[: (nz[-1] + 1 if nz.size else 0)]


class Evaluations:
    """fft/evaluations.rs:28-74: evaluations of a polynomial over a domain (thin wrappers over the transforms)."""

    def __init__(self, evaluations, domain):
        self.evaluations = np.ascontiguousarray(evaluations, dtype=np.uint64).reshape(-1, 4)
        self.domain = domain

    @classmethod
    def from_vec_and_domain(cls, evaluations, domain):  # evaluations.rs:42-44
        return cls(evaluations, domain)

    def interpolate(self):  # evaluations.rs:52-56: iFFT, then trim trailing zeros (DensePolynomial::from_coefficients_vec)
        coeffs = self.domain.ifft(self.evaluations)
        nz = np.nonzero(coeffs.any(axis=1))[0]
        return coeffs[: (nz[-1] + 1 if nz.size else 0)]

    def interpolate_by_ref(self):  # evaluations.rs:58-62: the same without consuming self
        return self.interpolate()

    def interpolate_with_pc(self, pc=None):  # evaluations.rs:64-74: the precomputation is the device's own table
        return self.interpolate()


def evaluate_over_domain(coeffs, domain):
    """`Polynomial::evaluate_over_domain` for a dense polynomial (fft/polynomial/mod.rs:261-300).

    deg < |domain|: zero-pad + FFT.  deg >= |domain|: the reference transforms every chunk of |domain| coefficients and adds
    the evaluation vectors (mod.rs:277-285); the transform is linear, so the same field elements come out of ONE transform of
    the chunk-wise sum of the coefficients - which is the remainder of the polynomial modulo X^|domain| - 1, computed on the
    device by the class fold of `divide_by_vanishing_poly` (poly.hip.h fr_fold_vanishing_kernel)."""
    import ctypes

    from . import _lib, poly

    coeffs = poly.trim(coeffs)  # `degree()` looks at the trimmed polynomial (dense.rs:88-96)
    if coeffs.shape[0] > domain.size:
        n = coeffs.shape[0]
        q = np.zeros((n - domain.size, 4), dtype=np.uint64)
        folded = np.zeros((domain.size, 4), dtype=np.uint64)
        _lib.check(_lib.lib().snarkvm_hip_fr_divide_by_vanishing(ctypes.c_void_p(q.ctypes.data), ctypes.c_void_p(folded.ctypes.data), ctypes.c_void_p(coeffs.ctypes.data),
                                                                ctypes.c_size_t(n), ctypes.c_size_t(domain.size), ctypes.c_int(0)))
        coeffs = folded
    return Evaluations(domain.fft(coeffs), domain)
