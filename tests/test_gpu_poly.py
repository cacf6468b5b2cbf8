"""Parity of the prover-round polynomial kernels (snarkvm_amd/csrc/poly.hip.h through the C ABI) and of KZG10::open
against the oracle, on a real MI355X.  Bit-exact: Fr results are unique Montgomery residues."""
import ctypes

import numpy as np
import pytest

from oracle import cpu as oracle
from snarkvm_amd import _lib, fft, kzg10, poly, synthetic
from tests import util
from tests.test_gpu_parity import _srs

pytestmark = pytest.mark.gpu

ONE = None


def _rnd(n, seed):
    return oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, seed)) if n else np.zeros((0, 4), dtype=np.uint64)


def _one():
    return util.ints_to_fr_mont([1])


def _linear_divisor(point):
    minus = oracle.fr_op("neg", np.asarray(point, dtype=np.uint64).reshape(1, 4))[0]
    return [(0, minus), (1, _one()[0])]


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 1000, 1024, 1025, 32 * 32 + 1, 70001, 1 << 18])
def test_vec_ops(n):
    a, b, c = _rnd(n, 1), _rnd(n, 2), _rnd(n, 3)
    s = _rnd(1, 4)
    for op in ("add", "sub", "mul", "mul_sub"):
        assert np.array_equal(poly.vec_op(op, a, b, c), oracle.fr_vec_op(op, a, b, c)), op
    assert np.array_equal(poly.vec_op("scale", a, scalar=s), oracle.fr_vec_op("scale", a, s))
    assert np.array_equal(poly.vec_op("sub_scalar", a, scalar=s), oracle.fr_vec_op("sub_scalar", a, s))
    assert np.array_equal(poly.vec_op("axpy", a, b, scalar=s), oracle.fr_vec_op("axpy", a, b, s))
    assert np.array_equal(poly.vec_op("rsub_scalar", a, scalar=s), oracle.fr_op("neg", oracle.fr_vec_op("sub_scalar", a, s)))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 8, 9, 31, 32, 33, 64, 1023, 1024, 1025, 2047, 2048, 2049, 4097, 32 * 32 * 32 + 5, 100003, 1 << 18, 2048 * 2048 + 5])  # 2 048 per workgroup, 8 per thread: one, two and three levels
def test_divide_by_linear_and_evaluate(n):
    """polynomial / (X - z) (kzg10/mod.rs:213-236) and DensePolynomial::evaluate (dense.rs:98-114)."""
    a = _rnd(n, 10 + n)
    z = _rnd(1, 5)
    q, rem = poly.divide_by_linear(a, z)
    wq, wr = oracle.poly_divide(a, _linear_divisor(z))
    assert np.array_equal(q, wq)
    want_val = oracle.poly_evaluate(a, z)
    assert np.array_equal(rem, want_val)
    assert np.array_equal(poly.evaluate(a, z), want_val)
    if wr.shape[0]:
        assert np.array_equal(wr, want_val)  # remainder theorem


def test_divide_by_linear_special_points():
    a = _rnd(5000, 77)
    a[-3:] = 0  # untrimmed input
    for z in (np.zeros((1, 4), dtype=np.uint64), _one(), oracle.fr_op("neg", _one())):
        q, rem = poly.divide_by_linear(a, z)
        wq, _ = oracle.poly_divide(a, _linear_divisor(z))
        assert np.array_equal(q, wq)
        assert np.array_equal(rem, oracle.poly_evaluate(a, z))
    # exact division: (X - z) * q has remainder zero
    z = _rnd(1, 6)
    q = _rnd(3000, 8)
    prod = oracle.polymul(12, [q, np.stack([_linear_divisor(z)[0][1], _one()[0]])])
    got_q, rem = poly.divide_by_linear(prod, z)
    assert np.array_equal(got_q, q) and not rem.any()


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 1000, 4096, 70001, 1 << 18])
def test_batch_inversion_and_mul(n):
    v = _rnd(n, 20 + n)
    for z in (0, 5, 17, n - 1, n // 2):
        if n > 3 and z < n:
            v[z] = 0
    coeff = _rnd(1, 21)
    assert np.array_equal(poly.batch_inversion_and_mul(v, coeff), oracle.batch_inversion_and_mul(v, coeff))
    assert np.array_equal(poly.batch_inversion_and_mul(v, _one()), oracle.batch_inversion_and_mul(v, _one()))


def test_batch_inversion_all_zero_and_roundtrip():
    z = np.zeros((100, 4), dtype=np.uint64)
    assert not poly.batch_inversion_and_mul(z, _one()).any()
    v = _rnd(5000, 31)
    inv = poly.batch_inversion_and_mul(v, _one())
    assert np.array_equal(poly.vec_op("mul", v, inv), np.tile(_one(), (5000, 1)))


@pytest.mark.parametrize("n", [1, 31, 33, 5000, 1 << 16, (1 << 18) + 7])
def test_distribute_powers(n):
    v = _rnd(n, 40 + n)
    g, c = _rnd(1, 41), _rnd(1, 42)
    assert np.array_equal(poly.distribute_powers_and_mul_by_const(v, g, c), oracle.distribute_powers(v, g, c))


@pytest.mark.parametrize("lg", [0, 1, 2, 5, 6, 10, 16])
def test_evaluate_all_lagrange_coefficients(lg):
    tau = _rnd(1, 50 + lg)
    assert np.array_equal(poly.evaluate_all_lagrange_coefficients(1 << lg, tau), oracle.lagrange_coefficients(lg, tau))
    # tau in the domain: one-hot branch
    gen = oracle.domain(lg)[0:1]
    k = (1 << lg) * 3 // 4
    tau_in = oracle.distribute_powers(np.tile(_one(), ((1 << lg), 1)), gen, _one())[k : k + 1]
    got = poly.evaluate_all_lagrange_coefficients(1 << lg, tau_in)
    assert np.array_equal(got, oracle.lagrange_coefficients(lg, tau_in))
    assert np.array_equal(got[k], _one()[0]) and int(got.any(axis=1).sum()) == 1


@pytest.mark.parametrize("n,D", [(5, 8), (8, 8), (9, 8), (16, 8), (40, 8), (3 * 4096 + 17, 4096), (1 << 17, 1 << 16), ((1 << 17) + 5, 1 << 16)])
def test_divide_and_mul_by_vanishing(n, D):
    a = _rnd(n, 60 + n)
    q, r = poly.divide_by_vanishing_poly(a, D)
    wq, wr = oracle.poly_divide(a, [(0, oracle.fr_op("neg", _one())[0]), (D, _one()[0])])
    assert np.array_equal(q, wq) and np.array_equal(r, wr)
    m = poly.mul_by_vanishing_poly(a, D)
    assert np.array_equal(m, poly.trim(oracle.mul_by_vanishing(a, D)))
    q2, r2 = poly.divide_by_vanishing_poly(m, D)
    assert np.array_equal(q2, poly.trim(a)) and r2.shape[0] == 0


def test_varuna_h0_on_device(golden):
    """KAT-polymul16 end to end on the device: (iNTT(z_a) * iNTT(z_b) - iNTT(z_c)) / (X^8 - 1) == h_0.txt."""
    from snarkvm_amd import fft
    from tests.test_oracle import _kat_polymul16

    z_a, z_b, z_c, h_0 = _kat_polymul16(golden)
    dom = fft.EvaluationDomain.new(8)
    ca, cb, cc = (dom.ifft(util.ints_to_fr_mont(v)) for v in (z_a, z_b, z_c))
    m = fft.PolyMultiplier()
    m.add_polynomial(poly.trim(ca), "z_a")
    m.add_polynomial(poly.trim(cb), "z_b")
    rowcheck = m.multiply()
    cpad = np.zeros_like(rowcheck)
    cpad[: cc.shape[0]] = cc[: rowcheck.shape[0]]
    rowcheck = poly.vec_op("sub", rowcheck, cpad)
    q, r = poly.divide_by_vanishing_poly(rowcheck, 8)
    assert r.shape[0] == 0
    assert util.fr_mont_to_ints(q) == h_0


def test_kzg10_open_matches_reference_formula(golden):
    """KZG10::open (kzg10/mod.rs:304-322): w = commit(p / (X - z)) [+ commit_gamma(blinding / (X - z))], random_v =
    blinding(z); against the oracle's long division + CPU MSM."""
    n = 2500
    powers_g = _srs(golden, n)
    gamma_g = oracle.g1_gen_bases(util.g1_generator_affine(), 11, 8)
    pw = kzg10.Powers(powers_g, gamma_g)
    coeffs = _rnd(n, 4243)
    point = _rnd(1, 4244)
    blind = _rnd(3, 4245)
    for hiding in (False, True):
        rand = kzg10.KZGRandomness(blind) if hiding else kzg10.KZGRandomness.empty()
        proof = kzg10.KZG10.open(pw, coeffs, point, rand)
        wq, _ = oracle.poly_divide(coeffs, _linear_divisor(point))
        want = oracle.g1_msm(powers_g, oracle.fr_op("to_bigint", wq), oracle.MSM_BATCHED)
        if hiding:
            bq, _ = oracle.poly_divide(blind, _linear_divisor(point))
            want = oracle.g1_add(want, oracle.g1_msm(gamma_g, oracle.fr_op("to_bigint", bq), oracle.MSM_BATCHED))
            assert np.array_equal(proof.random_v, oracle.poly_evaluate(blind, point))
        else:
            assert proof.random_v is None
        assert util.affine_equal(np.array([proof.w]), oracle.g1_to_affine(want))
    with pytest.raises(kzg10.PCError):
        kzg10.KZG10.open(pw, _rnd(n + 1, 1), point, kzg10.KZGRandomness.empty())
    pw.close()


def test_kzg10_open_lagrange(golden):
    """KZG10::open_lagrange (kzg10/mod.rs:274-301) over a stand-in Lagrange basis."""
    lg = 9
    n = 1 << lg
    basis_g = _srs(golden, n)
    pw = kzg10.Powers(basis_g, oracle.g1_gen_bases(util.g1_generator_affine(), 3, 2))
    gen = oracle.domain(lg)[0:1]
    elems = oracle.distribute_powers(np.tile(_one(), (n, 1)), gen, _one())
    evals = _rnd(n, 9001)
    point, value = _rnd(1, 9002), _rnd(1, 9003)
    proof = kzg10.KZG10.open_lagrange(pw, elems, evals, point, value)
    div = oracle.batch_inversion_and_mul(oracle.fr_vec_op("sub_scalar", elems, point), _one())
    wit = oracle.fr_vec_op("mul", div, oracle.fr_vec_op("sub_scalar", evals, value))
    want = oracle.g1_msm(basis_g, oracle.fr_op("to_bigint", wit), oracle.MSM_BATCHED)
    assert util.affine_equal(np.array([proof.w]), oracle.g1_to_affine(want)) and proof.random_v is None
    with pytest.raises(kzg10.PCError):
        kzg10.KZG10.open_lagrange(pw, elems, evals, elems[5:6], value)
    pw.close()


def test_device_resident_round2_slice_and_openings(golden):
    """a Varuna second-round slice that never leaves HBM between its steps - evaluations -> device
    iNTT -> z_a * z_b - z_c on the 2|R| domain (device NTTs + rowcheck kernel) -> device iNTT -> division by X^|R| - 1
    -> commit straight from the device vector; then three openings (kzg10/mod.rs:304-322).  Every result is compared
    with the oracle's CPU restatement of the same reference functions."""
    import ctypes

    import torch

    lgR = 12
    n, n2 = 1 << lgR, 1 << (lgR + 1)
    L = _lib.lib()
    powers_g = _srs(golden, n2)
    gamma_g = oracle.g1_gen_bases(util.g1_generator_affine(), 7, 4)
    pw = kzg10.Powers(powers_g, gamma_g)
    dev = torch.device("cuda:0")

    def P(t):
        return ctypes.c_void_p(t.data_ptr())

    def to_dev(a, pad_to=None):
        t = torch.zeros((pad_to or a.shape[0], 4), dtype=torch.int64, device=dev)
        t[: a.shape[0]] = torch.from_numpy(a.view(np.int64)).to(dev)
        return t

    def ntt_dev(t, lg, direction):
        torch.cuda.synchronize()
        _lib.check(L.snarkvm_hip_ntt_device(P(t), ctypes.c_uint32(lg), 0, direction, 0))

    # z_a, z_b, z_c evaluations on R with z_a * z_b == z_c on R (so that the quotient is exact)
    ea, eb = _rnd(n, 501), _rnd(n, 502)
    ec = oracle.fr_vec_op("mul", ea, eb)
    d = [to_dev(e, n2) for e in (ea, eb, ec)]
    for t in d:
        ntt_dev(t, lgR, 1)          # coefficients of z_m (degree < |R|), zero-padded to 2|R|
    dc_coeffs = d[2].clone()
    for t in d[:2]:
        ntt_dev(t, lgR + 1, 0)      # evaluations on the doubled domain
    prod = torch.empty_like(d[0])
    torch.cuda.synchronize()
    _lib.check(L.snarkvm_hip_fr_vec_op(2, P(prod), P(d[0]), P(d[1]), None, None, ctypes.c_size_t(n2), 1))
    ntt_dev(prod, lgR + 1, 1)       # z_a * z_b in coefficient form
    _lib.check(L.snarkvm_hip_fr_vec_op(1, P(prod), P(prod), P(dc_coeffs), None, None, ctypes.c_size_t(n2), 1))  # - z_c
    quot = torch.zeros((n2 - n, 4), dtype=torch.int64, device=dev)
    rem = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    _lib.check(L.snarkvm_hip_fr_divide_by_vanishing(P(quot), P(rem), P(prod), ctypes.c_size_t(n2), ctypes.c_size_t(n), 1))
    assert not bool(rem.any())      # exact: z_a z_b - z_c vanishes on R
    comm, _ = kzg10.KZG10.commit_device(pw, quot)
    # oracle: the same pipeline with the restated reference functions
    ca, cb, cc = (oracle.ntt(e, oracle.ORDER_NN, oracle.INVERSE) for e in (ea, eb, ec))
    rowcheck = oracle.polymul(lgR + 1, [ca, cb])
    cpad = np.zeros_like(rowcheck)
    cpad[:n] = cc
    rowcheck = oracle.fr_vec_op("sub", rowcheck, cpad)
    h0, r0 = oracle.poly_divide(rowcheck, [(0, oracle.fr_op("neg", _one())[0]), (n, _one()[0])])
    assert r0.shape[0] == 0
    got_h0 = quot.cpu().numpy().view(np.uint64)
    assert np.array_equal(poly.trim(got_h0), h0)
    want = oracle.g1_msm(powers_g[: h0.shape[0]], oracle.fr_op("to_bigint", h0), oracle.MSM_BATCHED)
    assert util.affine_equal(oracle.g1_to_affine(comm), oracle.g1_to_affine(want))
    # hiding commit from the device vector equals the host-path commit
    blind = _rnd(3, 77)
    c_dev, _ = kzg10.KZG10.commit_device(pw, quot, 1, lambda k: blind[:k])
    c_host, _ = kzg10.KZG10.commit(pw, got_h0, 1, lambda k: blind[:k])
    assert util.affine_equal(oracle.g1_to_affine(c_dev), oracle.g1_to_affine(c_host))
    # three openings of the committed polynomials at one challenge point
    point = _rnd(1, 88)
    for coeffs in (ca, cb, h0):
        proof = kzg10.KZG10.open(pw, coeffs, point, kzg10.KZGRandomness.empty())
        wq, _ = oracle.poly_divide(coeffs, _linear_divisor(point))
        want = oracle.g1_msm(powers_g[: wq.shape[0]], oracle.fr_op("to_bigint", wq), oracle.MSM_BATCHED)
        assert util.affine_equal(np.array([proof.w]), oracle.g1_to_affine(want))
    pw.close()


@pytest.mark.parametrize("n,lg", [(5, 3), (8, 3), (9, 3), (1000, 8), (3 * 256 + 77, 8), (4 * 4096, 12), (4096 * 7 + 1, 12)])
def test_evaluate_over_domain_incl_long_polynomials(n, lg):
    """`Polynomial::evaluate_over_domain` (fft/polynomial/mod.rs:261-300): polynomials of degree >= |domain| are transformed
    chunk by chunk and the evaluation vectors added (mod.rs:277-285) - restated here literally with the oracle's NTT and vector
    addition; the device folds the coefficients first (remainder mod X^|D| - 1) and transforms once: same field elements."""
    from snarkvm_amd import fft

    D = 1 << lg
    coeffs = oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, 8100 + n))
    coeffs[-1, :] = 0  # a trailing zero coefficient: `degree()` is taken after trimming
    coeffs[-1, 0] = 0
    dom = fft.EvaluationDomain.new(D)
    got = fft.evaluate_over_domain(coeffs, dom).evaluations
    trimmed = coeffs[:-1]
    if trimmed.shape[0] > D:
        acc = np.zeros((D, 4), dtype=np.uint64)
        for lo in range(0, trimmed.shape[0], D):
            chunk = np.zeros((D, 4), dtype=np.uint64)
            part = trimmed[lo : lo + D]
            chunk[: part.shape[0]] = part
            acc = oracle.fr_vec_op("add", acc, oracle.ntt(chunk))
        want = acc
    else:
        pad = np.zeros((D, 4), dtype=np.uint64)
        pad[: trimmed.shape[0]] = trimmed
        want = oracle.ntt(pad)
    assert np.array_equal(got, want)


def test_element_wise_arithmetic_4_over_domain():
    """`PolyMultiplier::element_wise_arithmetic_4_over_domain` (multiplier.rs:136-178): two polynomials and two evaluation vectors
    on a 2^10 domain combined by f = a * b - c + d on the device, against the same composition of the oracle's transforms."""
    lg = 10
    n = 1 << lg
    dom = fft.EvaluationDomain.new(n)
    pa, pb = _rnd(300, 501), _rnd(n, 502)  # a short and a full-length polynomial
    ec, ed = _rnd(n, 503), _rnd(700, 504)  # a full and a short (zero-padded) evaluation vector
    m = fft.PolyMultiplier()
    m.add_polynomial(pa, "a")
    m.add_polynomial(pb, "b")
    m.add_evaluation(ec, "c")
    m.add_evaluation(ed, "d")
    got = m.element_wise_arithmetic_4_over_domain(dom, ["a", "b", "c", "d"], lambda a, b, c, d: poly.vec_op("add", poly.vec_op("mul_sub", a, b, c), d))

    def pad(v):
        out = np.zeros((n, 4), dtype=np.uint64)
        out[: v.shape[0]] = v
        return out

    ea, eb = oracle.ntt(pad(pa)), oracle.ntt(pad(pb))
    comb = oracle.fr_op("add", oracle.fr_op("sub", oracle.fr_op("mul", ea, eb), ec), pad(ed))
    want = poly.trim(oracle.ntt(comb, oracle.ORDER_NN, oracle.INVERSE, oracle.STANDARD))
    assert np.array_equal(got, want)
    with pytest.raises(AssertionError):
        m2 = fft.PolyMultiplier()
        m2.add_polynomial(pa, "a")
        m2.element_wise_arithmetic_4_over_domain(dom, ["a", "a", "a", "a"], lambda a, b, c, d: a)


@pytest.mark.parametrize("lg", [0, 1, 3, 10, 16])
def test_domain_constants_and_roots_of_unity(lg, golden):
    """EvaluationDomain's fields (domain.rs:83-147) and `roots_of_unity` / `precompute_fft` / `precompute_ifft` (domain.rs:360-372,
    594-648): group_gen has order exactly 2^lg, size_inv * size = 1, generator_inv * 22 = 1, the roots are the successive powers."""
    dom = fft.EvaluationDomain.new(1 << lg)
    one = _one()
    mul = lambda a, b: oracle.fr_op("mul", a, b)  # noqa: E731
    assert np.array_equal(mul(dom.size_inv, dom.size_as_field_element), one)
    assert np.array_equal(mul(dom.group_gen, dom.group_gen_inv), one)
    assert np.array_equal(mul(dom.generator_inv, fft._fr(22)), one)
    g = dom.group_gen
    for _ in range(lg):
        last = g
        g = mul(g, g)
    assert np.array_equal(g, one) and (lg == 0 or not np.array_equal(last, one))  # order exactly 2^lg
    if lg == 3:  # the reference's own size-8 domain (tests/golden: resources/circuit_0/domain/R.txt)
        assert golden is not None
    pc, ipc = dom.precompute_fft(), dom.precompute_ifft()
    assert pc.roots.shape[0] == (1 << lg) // 2 and ipc.inverse and pc.to_ifft_precomputation().roots.shape == ipc.roots.shape
    for roots, w in ((pc.roots, dom.group_gen), (ipc.roots, dom.group_gen_inv)):
        if roots.shape[0]:
            assert np.array_equal(roots[0:1], one)
            assert np.array_equal(roots[1:], oracle.fr_op("mul", roots[:-1], np.repeat(w, roots.shape[0] - 1, axis=0)))
            assert np.array_equal(mul(roots[-1:], w), oracle.fr_op("neg", one))  # w^(n/2) = -1
    x = _rnd(1 << lg, 900 + lg)
    assert np.array_equal(dom.fft_with_pc(x, pc), dom.fft(x)) and np.array_equal(dom.ifft_with_pc(x, ipc), dom.ifft(x))
    ev = fft.Evaluations.from_vec_and_domain(x, dom)
    assert np.array_equal(ev.interpolate_by_ref(), ev.interpolate()) and np.array_equal(ev.interpolate_with_pc(ipc), ev.interpolate())
