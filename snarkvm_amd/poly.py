"""Prover-round polynomial operations on the gfx950 backend (SURVEY.md §8f N2 and the `open` half of row a15).

Host mirror of the reference functions the Varuna prover runs between its NTTs and its commitments:

    DensePolynomial::{evaluate, mul_by_vanishing_poly, divide_by_vanishing_poly}   fft/polynomial/dense.rs:98-114, 153-169
    `polynomial / (X - point)` (Polynomial::divide_with_q_and_r)                    fft/polynomial/mod.rs:222-256
    batch_inversion / batch_inversion_and_mul                                       fields/src/lib.rs:66-129
    EvaluationDomain::{distribute_powers_and_mul_by_const,
                       evaluate_all_lagrange_coefficients}                          fft/domain.rs:224-292

Vectors are (n, 4) u64 arrays of Montgomery limbs (the Rust `Vec<Fr>` memory image); scalars are (4,) / (1, 4).
Every function runs on the device - there is no CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib

VEC_OPS = {"add": 0, "sub": 1, "mul": 2, "mul_sub": 3, "scale": 4, "sub_scalar": 5, "axpy": 6, "rsub_scalar": 7}


def _v(x):
    return np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p()


def trim(coeffs):
    """DensePolynomial::from_coefficients_vec (dense.rs:76-86): drop trailing zero coefficients."""
    coeffs = _v(coeffs)
    nz = np.nonzero(coeffs.any(axis=1))[0]
    return coeffs[: (int(nz[-1]) + 1 if nz.size else 0)]


def vec_op(op, a, b=None, c=None, scalar=None):
    """Element-wise Fr arithmetic: add, sub, mul, mul_sub (a*b-c), scale (a*s), sub_scalar (a-s), axpy (a+b*s), rsub_scalar (s-a)."""
    a = _v(a)
    n = a.shape[0]
    b = None if b is None else _v(b)
    c = None if c is None else _v(c)
    for other in (b, c):
        if other is not None and other.shape[0] != n:
            raise ValueError("length mismatch")  # the reference zips with zip_eq
    s = None if scalar is None else _v(scalar)
    out = np.empty_like(a)
    _lib.check(_lib.lib().snarkvm_hip_fr_vec_op(ctypes.c_int(VEC_OPS[op]), _p(out), _p(a), _p(b), _p(c), _p(s), ctypes.c_size_t(n),
                                               ctypes.c_int(0)))
    return out


def divide_by_linear(coeffs, point):
    """(quotient, p(point)) of p / (X - point); the quotient is trimmed like a DensePolynomial."""
    coeffs = trim(coeffs)
    n = coeffs.shape[0]
    q = np.zeros((max(n - 1, 0), 4), dtype=np.uint64)
    rem = np.zeros((1, 4), dtype=np.uint64)
    _lib.check(_lib.lib().snarkvm_hip_fr_divide_by_linear(_p(q) if n > 1 else ctypes.c_void_p(), _p(rem), _p(coeffs), ctypes.c_size_t(n),
                                                         _p(_v(point)), ctypes.c_int(0)))
    return trim(q), rem


def evaluate(coeffs, point):
    """DensePolynomial::evaluate (dense.rs:98-114)."""
    coeffs = trim(coeffs)
    rem = np.zeros((1, 4), dtype=np.uint64)
    _lib.check(_lib.lib().snarkvm_hip_fr_divide_by_linear(ctypes.c_void_p(), _p(rem), _p(coeffs), ctypes.c_size_t(coeffs.shape[0]),
                                                         _p(_v(point)), ctypes.c_int(0)))
    return rem


def batch_inversion_and_mul(v, coeff):
    """fields/src/lib.rs:73-129: returns coeff / v_i (zeros stay zero)."""
    v = np.array(v, dtype=np.uint64, copy=True).reshape(-1, 4)
    _lib.check(_lib.lib().snarkvm_hip_fr_batch_inversion_and_mul(_p(v), ctypes.c_size_t(v.shape[0]), _p(_v(coeff)), ctypes.c_int(0)))
    return v


def distribute_powers_and_mul_by_const(v, g, c):
    """fft/domain.rs:229-254: v_i * c * g^i."""
    v = np.array(v, dtype=np.uint64, copy=True).reshape(-1, 4)
    _lib.check(_lib.lib().snarkvm_hip_fr_distribute_powers(_p(v), ctypes.c_size_t(v.shape[0]), _p(_v(g)), _p(_v(c)), ctypes.c_int(0)))
    return v


def evaluate_all_lagrange_coefficients(domain_size, tau):
    """fft/domain.rs:258-292 for the power-of-two domain of `domain_size` elements."""
    lg = domain_size.bit_length() - 1
    if domain_size <= 0 or 1 << lg != domain_size:
        raise ValueError("domain_size is not power of 2")
    out = np.zeros((domain_size, 4), dtype=np.uint64)
    _lib.check(_lib.lib().snarkvm_hip_fr_lagrange_coefficients(_p(out), ctypes.c_uint32(lg), _p(_v(tau)), ctypes.c_int(0)))
    return out


def divide_by_vanishing_poly(coeffs, domain_size):
    """dense.rs:161-169: (quotient, remainder) of p / (X^domain_size - 1), both trimmed."""
    coeffs = trim(coeffs)
    n = coeffs.shape[0]
    if n == 0:
        return coeffs, coeffs
    q = np.zeros((max(n - domain_size, 0), 4), dtype=np.uint64)
    r = np.zeros((min(n, domain_size), 4), dtype=np.uint64)
    _lib.check(_lib.lib().snarkvm_hip_fr_divide_by_vanishing(_p(q) if q.shape[0] else ctypes.c_void_p(), _p(r), _p(coeffs), ctypes.c_size_t(n),
                                                            ctypes.c_size_t(domain_size), ctypes.c_int(0)))
    return trim(q), trim(r)


def mul_by_vanishing_poly(coeffs, domain_size):
    """dense.rs:153-159: p * (X^domain_size - 1), trimmed."""
    coeffs = _v(coeffs)
    out = np.zeros((coeffs.shape[0] + domain_size, 4), dtype=np.uint64)
    _lib.check(_lib.lib().snarkvm_hip_fr_mul_by_vanishing(_p(out), _p(coeffs) if coeffs.shape[0] else ctypes.c_void_p(),
                                                         ctypes.c_size_t(coeffs.shape[0]), ctypes.c_size_t(domain_size), ctypes.c_int(0)))
    return trim(out)
