"""The oracle's prover-round polynomial helpers (oracle/cpu_oracle.cpp, restating fft/polynomial/{mod,dense}.rs,
fields/src/lib.rs, fft/domain.rs) pinned against independent Python big-int arithmetic and the Varuna h_0 vector."""
import random

import numpy as np
import pytest

from oracle import cpu as oracle
from oracle import pyref
from tests import util
from tests.test_oracle import _kat_polymul16

P = pyref.R_MOD


def _rand(n, seed, zeros=()):
    rng = random.Random(seed)
    v = [rng.randrange(P) for _ in range(n)]
    for z in zeros:
        if z < n:
            v[z] = 0
    return v


def _py_divide(a, divisor):
    """schoolbook long division; divisor = dense coefficient list with non-zero leading coefficient"""
    a = list(a)
    while a and a[-1] == 0:
        a.pop()
    d = len(divisor) - 1
    if len(a) - 1 < d:
        return [], a
    q = [0] * (len(a) - d)
    inv = pow(divisor[-1], P - 2, P)
    for i in range(len(a) - 1, d - 1, -1):
        c = a[i] * inv % P
        q[i - d] = c
        for j, dc in enumerate(divisor):
            a[i - d + j] = (a[i - d + j] - c * dc) % P
    r = a[:d]
    while r and r[-1] == 0:
        r.pop()
    while q and q[-1] == 0:
        q.pop()
    return q, r


@pytest.mark.parametrize("n", [0, 1, 2, 3, 33, 200])
def test_divide_by_linear_and_evaluate(n):
    a = _rand(n, 100 + n)
    if n > 2:
        a[-1] = 0  # an untrimmed input: the reference's DensePolynomial would have dropped it
    z = _rand(1, 7)[0]
    q, r = oracle.poly_divide(util.ints_to_fr_mont(a), [(0, util.ints_to_fr_mont([-z % P])[0]), (1, util.ints_to_fr_mont([1])[0])])
    wq, wr = _py_divide(a, [-z % P, 1])
    assert util.fr_mont_to_ints(q) == wq and util.fr_mont_to_ints(r) == wr
    val = util.fr_mont_to_ints(oracle.poly_evaluate(util.ints_to_fr_mont(a), util.ints_to_fr_mont([z])))[0]
    assert val == pyref.horner(a, z) % P
    assert (wr[0] if wr else 0) == val  # remainder theorem


def test_divide_by_vanishing_reproduces_h0(golden):
    """(iNTT(z_a) * iNTT(z_b) - iNTT(z_c)) / (X^8 - 1) == h_0.txt with zero remainder, through the oracle's long division."""
    z_a, z_b, z_c, h_0 = _kat_polymul16(golden)
    ca, cb, cc = (oracle.ntt(util.ints_to_fr_mont(v), oracle.ORDER_NN, oracle.INVERSE) for v in (z_a, z_b, z_c))
    prod = oracle.polymul(4, [ca, cb])
    cpad = np.zeros_like(prod)
    cpad[:8] = cc
    diff = oracle.fr_vec_op("sub", prod, cpad)
    one = util.ints_to_fr_mont([1])[0]
    q, r = oracle.poly_divide(diff, [(0, util.ints_to_fr_mont([P - 1])[0]), (8, one)])
    assert r.shape[0] == 0
    assert util.fr_mont_to_ints(q) == h_0
    # and multiplying back
    back = oracle.mul_by_vanishing(q, 8)
    assert util.fr_mont_to_ints(back)[: diff.shape[0]] == util.fr_mont_to_ints(diff)[: back.shape[0]]


@pytest.mark.parametrize("n,D", [(5, 8), (8, 8), (9, 8), (40, 8), (100, 32)])
def test_divide_by_vanishing_general(n, D):
    a = _rand(n, 5 * n + D)
    one = util.ints_to_fr_mont([1])[0]
    q, r = oracle.poly_divide(util.ints_to_fr_mont(a), [(0, util.ints_to_fr_mont([P - 1])[0]), (D, one)])
    wq, wr = _py_divide(a, [P - 1] + [0] * (D - 1) + [1])
    assert util.fr_mont_to_ints(q) == wq and util.fr_mont_to_ints(r) == wr


def test_batch_inversion_and_mul_skips_zeros():
    v = _rand(50, 3, zeros=(0, 7, 8, 49))
    c = _rand(1, 4)[0]
    got = util.fr_mont_to_ints(oracle.batch_inversion_and_mul(util.ints_to_fr_mont(v), util.ints_to_fr_mont([c])))
    assert got == [(c * pow(x, P - 2, P)) % P if x else 0 for x in v]
    allz = oracle.batch_inversion_and_mul(np.zeros((4, 4), dtype=np.uint64), util.ints_to_fr_mont([c]))
    assert not allz.any()


def test_vec_ops_and_distribute_powers():
    a, b, c = _rand(20, 1), _rand(20, 2), _rand(20, 3)
    A, B, C = (util.ints_to_fr_mont(x) for x in (a, b, c))
    s = _rand(1, 9)[0]
    S = util.ints_to_fr_mont([s])
    want = {
        "add": [(x + y) % P for x, y in zip(a, b)], "sub": [(x - y) % P for x, y in zip(a, b)],
        "mul": [x * y % P for x, y in zip(a, b)], "mul_sub": [(x * y - z) % P for x, y, z in zip(a, b, c)],
    }
    for op, w in want.items():
        assert util.fr_mont_to_ints(oracle.fr_vec_op(op, A, B, C)) == w, op
    assert util.fr_mont_to_ints(oracle.fr_vec_op("scale", A, S)) == [x * s % P for x in a]
    assert util.fr_mont_to_ints(oracle.fr_vec_op("sub_scalar", A, S)) == [(x - s) % P for x in a]
    assert util.fr_mont_to_ints(oracle.fr_vec_op("axpy", A, B, S)) == [(x + y * s) % P for x, y in zip(a, b)]
    g = _rand(1, 11)[0]
    got = util.fr_mont_to_ints(oracle.distribute_powers(A, util.ints_to_fr_mont([g]), S))
    assert got == [x * s * pow(g, i, P) % P for i, x in enumerate(a)]


@pytest.mark.parametrize("lg", [0, 1, 3, 6])
def test_lagrange_coefficients(lg):
    n = 1 << lg
    w = pyref.domain_group_gen(lg)
    tau = _rand(1, 77 + lg)[0]
    got = util.fr_mont_to_ints(oracle.lagrange_coefficients(lg, util.ints_to_fr_mont([tau])))
    # definition: L_i(tau) = prod_{j != i} (tau - w^j) / (w^i - w^j)
    want = []
    for i in range(n):
        num = den = 1
        for j in range(n):
            if j != i:
                num = num * (tau - pow(w, j, P)) % P
                den = den * (pow(w, i, P) - pow(w, j, P)) % P
        want.append(num * pow(den, P - 2, P) % P)
    assert got == want
    # tau inside the domain: the one-hot branch (domain.rs:265-275)
    k = n // 2
    got = util.fr_mont_to_ints(oracle.lagrange_coefficients(lg, util.ints_to_fr_mont([pow(w, k, P)])))
    assert got == [1 if i == k else 0 for i in range(n)]
