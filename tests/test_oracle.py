"""Pins the oracles (oracle/pyref.py and oracle/cpu_oracle.cpp) against the reference's own
known answers (tests/golden/, extracted from the reference by tests/golden/make_golden.py) and
against the reference's differential properties.  CPU only."""
import numpy as np
import pytest

from oracle import cpu as oracle
from oracle import pyref
from snarkvm_amd import synthetic
from tests import util

L = pyref.from_limbs


# ------------------------------------------------------------------ field constants (KAT 1)
def test_field_constants_match_reference(golden):
    fr, fq = golden["constants"]["fr"], golden["constants"]["fq"]
    assert L(fr["MODULUS"]) == pyref.R_MOD and L(fq["MODULUS"]) == pyref.Q_MOD
    assert L(fr["R"]) == pyref.FR_MONT_R and L(fq["R"]) == pyref.FQ_MONT_R
    assert L(fr["R2"]) == pow(2, 512, pyref.R_MOD) and L(fq["R2"]) == pow(2, 768, pyref.Q_MOD)
    assert fr["INV"] == (-pow(pyref.R_MOD, -1, 1 << 64)) % (1 << 64)
    assert fq["INV"] == (-pow(pyref.Q_MOD, -1, 1 << 64)) % (1 << 64)
    assert L(fr["GENERATOR"]) == pyref.fr_to_mont(pyref.FR_GENERATOR)
    assert L(fr["TWO_ADIC_ROOT_OF_UNITY"]) == pyref.fr_to_mont(pyref.FR_TWO_ADIC_ROOT)
    assert fr["TWO_ADICITY"] == pyref.FR_TWO_ADICITY and fr["MODULUS_BITS"] == 253 and fr["REPR_SHAVE_BITS"] == 3
    # C++ oracle: from_bigint(1) == R, to_bigint(R) == 1, R2 via mul
    one = util.limbs(1, 4)
    assert L(oracle.fr_op("from_bigint", one)[0]) == L(fr["R"])
    assert L(oracle.fr_op("to_bigint", np.array(fr["R"], dtype=np.uint64))[0]) == 1
    assert L(oracle.fq_op("from_bigint", util.limbs(1, 6))[0]) == L(fq["R"])
    assert L(oracle.fq_op("to_bigint", np.array(fq["R"], dtype=np.uint64))[0]) == 1


def test_two_adic_roots_table(golden):
    """curves/src/bls12_377/fr.rs:212-239 (test_powers_of_root_of_unity / test_two_adic_root_of_unity)."""
    tab = golden["constants"]["fr"]["POWERS_OF_ROOTS_OF_UNITY"]
    assert len(tab) == 46
    cur = np.array(golden["constants"]["fr"]["TWO_ADIC_ROOT_OF_UNITY"], dtype=np.uint64).reshape(1, 4)
    w = pyref.FR_TWO_ADIC_ROOT
    for i, entry in enumerate(tab):
        assert L(entry) == L(cur[0]) == pyref.fr_to_mont(w)
        cur = oracle.fr_op("sqr", cur)
        w = w * w % pyref.R_MOD
    # after 46 squarings of the table's first entry we are at w^(2^46) = -1, one more gives 1
    assert w == pyref.R_MOD - 1
    assert pow(pyref.FR_TWO_ADIC_ROOT, 1 << 47, pyref.R_MOD) == 1


def test_field_ops_vs_python():
    rng = np.random.default_rng(1)
    for mod, nl, op_fn, mont in ((pyref.R_MOD, 4, oracle.fr_op, pyref.FR_MONT_R), (pyref.Q_MOD, 6, oracle.fq_op, pyref.FQ_MONT_R)):
        vals_a = [int.from_bytes(rng.bytes(48), "little") % mod for _ in range(64)] + [0, 1, mod - 1]
        vals_b = [int.from_bytes(rng.bytes(48), "little") % mod for _ in range(64)] + [mod - 1, 0, mod - 1]
        a = np.array([pyref.to_limbs(v, nl) for v in vals_a], dtype=np.uint64)
        b = np.array([pyref.to_limbs(v, nl) for v in vals_b], dtype=np.uint64)
        rinv = pow(mont, -1, mod)
        got = {op: [L(r) for r in op_fn(op, a, b)] for op in ("add", "sub", "mul", "neg", "sqr", "from_bigint", "to_bigint")}
        for i, (x, y) in enumerate(zip(vals_a, vals_b)):
            assert got["add"][i] == (x + y) % mod
            assert got["sub"][i] == (x - y) % mod
            assert got["mul"][i] == x * y * rinv % mod
            assert got["sqr"][i] == x * x * rinv % mod
            assert got["neg"][i] == (-x) % mod
            assert got["from_bigint"][i] == x * mont % mod
            assert got["to_bigint"][i] == x * rinv % mod
        nz = a[:64]
        inv = op_fn("inverse", nz)
        for i in range(64):
            # Montgomery inverse: inv(aR) = a^-1 R
            assert L(inv[i]) * vals_a[i] % mod == mont * mont % mod


# ------------------------------------------------------------------ domain + NTT KATs (KAT 2, 3)
def test_domain_size8_matches_varuna_vectors(golden):
    dom = [int(x) for x in golden["varuna"]["domain"]["R"]]
    w = pyref.domain_group_gen(3)
    assert dom == [pow(w, i, pyref.R_MOD) for i in range(8)]
    assert golden["varuna"]["domain"]["C"] == golden["varuna"]["domain"]["R"]
    d = oracle.domain(3)
    assert pyref.fr_from_mont(L(d[0])) == w
    assert pyref.fr_from_mont(L(d[1])) == pow(w, -1, pyref.R_MOD)
    assert pyref.fr_from_mont(L(d[2])) == pow(8, -1, pyref.R_MOD)
    assert pyref.fr_from_mont(L(d[3])) == pow(22, -1, pyref.R_MOD)


def test_host_mirror_domain_constants_match_reference(golden):
    """snarkvm_amd.fft.EvaluationDomain's fields (domain.rs:83-147) against the reference's own size-8 domain vector
    (resources/circuit_0/domain/R.txt = the powers of group_gen) and the oracle's domain constants at several sizes."""
    from snarkvm_amd import fft

    dom = fft.EvaluationDomain.new(8)
    assert pyref.fr_from_mont(L(dom.group_gen[0])) == int(golden["varuna"]["domain"]["R"][1])
    for lg in (0, 1, 3, 16, 24, 47):
        dm = fft.EvaluationDomain.new(1 << lg)
        d = oracle.domain(lg)
        assert dm.log_size_of_group == lg and dm.size == 1 << lg
        assert np.array_equal(dm.group_gen[0], d[0]) and np.array_equal(dm.group_gen_inv[0], d[1])
        assert np.array_equal(dm.size_inv[0], d[2]) and np.array_equal(dm.generator_inv[0], d[3])
        assert pyref.fr_from_mont(L(dm.size_as_field_element[0])) == (1 << lg) % pyref.R_MOD
    assert fft.EvaluationDomain.new((1 << 47) + 1) is None  # domain.rs:118-147: no subgroup of that size


def _kat_intt8(golden):
    # public inputs 1,8,32,128 interleaved with private 2,4,2,0
    evals = [1, 2, 8, 4, 32, 2, 128, 0]
    z_lde = [int(x) for x in golden["varuna"]["polynomials"]["z_lde"]]
    return evals, z_lde


def test_kat_intt8_python(golden):
    evals, z_lde = _kat_intt8(golden)
    assert pyref.ntt(evals, inverse=True) == z_lde
    # and evaluating z_lde over domain C returns the evaluations
    dom = [int(x) for x in golden["varuna"]["domain"]["C"]]
    assert [pyref.horner(z_lde, x) for x in dom] == evals


def test_kat_intt8_cpp(golden):
    evals, z_lde = _kat_intt8(golden)
    got = oracle.ntt(util.ints_to_fr_mont(evals), oracle.ORDER_NN, oracle.INVERSE, oracle.STANDARD)
    assert util.fr_mont_to_ints(got) == z_lde
    back = oracle.ntt(got, oracle.ORDER_NN, oracle.FORWARD, oracle.STANDARD)
    assert util.fr_mont_to_ints(back) == evals


def _kat_polymul16(golden):
    z_a = [2, 2, 2, 2, 2, 8, 32, 0]
    z_b = [4, 4, 4, 4, 4, 4, 4, 0]
    z_c = [8, 8, 8, 8, 8, 32, 128, 0]
    h_0 = [int(x) for x in golden["varuna"]["polynomials"]["h_0"]]
    return z_a, z_b, z_c, h_0


def _divide_by_vanishing(poly, n):
    """exact division by X^n - 1 (algorithms/src/fft/polynomial/dense.rs:153-170 semantics)"""
    p = list(poly)
    q = [0] * max(0, len(p) - n)
    for i in range(len(p) - 1, n - 1, -1):
        q[i - n] = p[i]
        p[i - n] = (p[i - n] + p[i]) % pyref.R_MOD
        p[i] = 0
    return q, p[:n]


def test_kat_polymul16(golden):
    """(iNTT(z_a)*iNTT(z_b) - iNTT(z_c)) / (X^8 - 1) == h_0  (round_functions/second.rs:104-122)."""
    z_a, z_b, z_c, h_0 = _kat_polymul16(golden)
    pa, pb, pc = (pyref.ntt(v, inverse=True) for v in (z_a, z_b, z_c))
    for impl in ("python", "cpp"):
        if impl == "python":
            prod = pyref.poly_mul_naive(pa, pb)
        else:
            ca = oracle.ntt(util.ints_to_fr_mont(z_a), oracle.ORDER_NN, oracle.INVERSE)
            cb = oracle.ntt(util.ints_to_fr_mont(z_b), oracle.ORDER_NN, oracle.INVERSE)
            assert util.fr_mont_to_ints(ca) == pa
            prod = util.fr_mont_to_ints(oracle.polymul(4, [ca, cb]))
        prod = prod + [0] * (16 - len(prod))
        diff = [(x - (pc[i] if i < 8 else 0)) % pyref.R_MOD for i, x in enumerate(prod)]
        q, rem = _divide_by_vanishing(diff, 8)
        assert all(r == 0 for r in rem)
        while q and q[-1] == 0:
            q.pop()
        assert q == h_0, impl


# ------------------------------------------------------------------ NTT properties (fft/tests.rs)
@pytest.mark.parametrize("lg", range(0, 7))
def test_ntt_cpp_vs_definition(lg):
    n = 1 << lg
    vals = [int(x) for x in np.random.default_rng(lg).integers(0, 2**62, n)]
    vals = [(v * 0x123456789ABCDEF123456789 + 7) % pyref.R_MOD for v in vals]
    x = util.ints_to_fr_mont(vals)
    for inverse in (False, True):
        for coset in (False, True):
            want = pyref.ntt(vals, inverse=inverse, coset=coset)
            d = oracle.INVERSE if inverse else oracle.FORWARD
            t = oracle.COSET if coset else oracle.STANDARD
            nn = util.fr_mont_to_ints(oracle.ntt(x, oracle.ORDER_NN, d, t))
            assert nn == want
            nr = util.fr_mont_to_ints(oracle.ntt(x, oracle.ORDER_NR, d, t))
            assert nr == pyref.bitrev_permute(want)
            xr = util.ints_to_fr_mont(pyref.bitrev_permute(vals))
            rn = util.fr_mont_to_ints(oracle.ntt(xr, oracle.ORDER_RN, d, t))
            assert rn == want
            rr = util.fr_mont_to_ints(oracle.ntt(xr, oracle.ORDER_RR, d, t))
            assert rr == pyref.bitrev_permute(want)


def test_ntt_is_horner_evaluation():
    """fft/tests.rs:120-149 (test_fft_correctness): degree-31 poly, domains 2^5 and 2^6, plain and coset."""
    coeffs = [int(v) for v in synthetic.splitmix64(99, 32)]
    for lg in (5, 6):
        n = 1 << lg
        w = pyref.domain_group_gen(lg)
        padded = coeffs + [0] * (n - 32)
        ev = util.fr_mont_to_ints(oracle.ntt(util.ints_to_fr_mont(padded)))
        assert ev == [pyref.horner(coeffs, pow(w, i, pyref.R_MOD)) for i in range(n)]
        evc = util.fr_mont_to_ints(oracle.ntt(util.ints_to_fr_mont(padded), kind=oracle.COSET))
        assert evc == [pyref.horner(coeffs, 22 * pow(w, i, pyref.R_MOD) % pyref.R_MOD) for i in range(n)]
        rt = oracle.ntt(oracle.ntt(util.ints_to_fr_mont(padded), kind=oracle.COSET), direction=oracle.INVERSE, kind=oracle.COSET)
        assert util.fr_mont_to_ints(rt) == [c % pyref.R_MOD for c in padded]


@pytest.mark.parametrize("lg", [10, 14, 17])
def test_ntt_roundtrip_and_linearity_large(lg):
    n = 1 << lg
    a = oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, 11))
    b = oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, 12))
    fa, fb = oracle.ntt(a), oracle.ntt(b)
    assert np.array_equal(oracle.ntt(fa, direction=oracle.INVERSE), a)
    assert np.array_equal(oracle.ntt(oracle.fr_op("add", a, b)), oracle.fr_op("add", fa, fb))
    # NR followed by RN-inverse is the identity without any bit-reversal pass (polymul path)
    assert np.array_equal(oracle.ntt(oracle.ntt(a, oracle.ORDER_NR), oracle.ORDER_RN, oracle.INVERSE), a)


def test_polymul_vs_schoolbook():
    """fft/polynomial/dense.rs:629-690 (mul_polynomials_random / _n_random)."""
    rng = np.random.default_rng(5)
    for la, lb in [(1, 1), (3, 5), (17, 40), (64, 64), (70, 1)]:
        a = [int(v) % pyref.R_MOD for v in rng.integers(0, 2**63, la)]
        b = [int(v) % pyref.R_MOD for v in rng.integers(0, 2**63, lb)]
        lg = max(0, (la + lb - 1).bit_length()) if (la + lb) > 1 else 0
        while (1 << lg) < la + lb:
            lg += 1
        got = util.fr_mont_to_ints(oracle.polymul(lg, [util.ints_to_fr_mont(a), util.ints_to_fr_mont(b)]))
        want = pyref.poly_mul_naive(a, b)
        assert got[: len(want)] == want and all(v == 0 for v in got[len(want) :])
    # three coefficient-form factors and one evaluation-form factor
    a, b, c = ([int(v) for v in rng.integers(1, 2**40, k)] for k in (5, 6, 4))
    lg = 4  # 5 + 6 + 4 = 15 -> 16
    e_coeffs = [3, 1]  # evaluation-form factor (does not count towards the domain, multiplier.rs:74-75)
    e = oracle.ntt(util.ints_to_fr_mont(e_coeffs + [0] * 14))
    got = util.fr_mont_to_ints(oracle.polymul(lg, [util.ints_to_fr_mont(v) for v in (a, b, c)], [e]))
    want = pyref.poly_mul_naive(pyref.poly_mul_naive(pyref.poly_mul_naive(a, b), c), e_coeffs)
    want = want + [0] * (32 - len(want))
    want = [(want[i] + want[i + 16]) % pyref.R_MOD for i in range(16)]  # mod X^16 - 1
    assert got == want


# ------------------------------------------------------------------ curve + MSM
def test_generator_and_srs_points(golden):
    g1 = golden["constants"]["g1"]
    assert (int(g1["GENERATOR_X_DEC"]), int(g1["GENERATOR_Y_DEC"])) == pyref.G1_GEN
    assert L(g1["GENERATOR_X_MONT"]) == pyref.fq_to_mont(pyref.G1_GEN[0])
    assert L(g1["GENERATOR_Y_MONT"]) == pyref.fq_to_mont(pyref.G1_GEN[1])
    assert L(g1["WEIERSTRASS_B_MONT"]) == pyref.fq_to_mont(1)
    assert pyref.g1_is_on_curve(pyref.G1_GEN)
    assert pyref.g1_mul(pyref.G1_GEN, pyref.R_MOD) is None  # in the prime-order subgroup
    pts = util.srs_points_ints(golden["srs_g1"])
    assert len(pts) == 1024 and pts[0] == pyref.G1_GEN
    assert all(pyref.g1_is_on_curve(p) for p in pts[:64])
    aff = util.g1_affine_from_ints(pts)
    assert oracle.g1_is_on_curve(aff)
    # G2 constants
    g2 = golden["constants"]["g2"]
    assert L(golden["constants"]["fq2"]["NONRESIDUE_MONT"]) == pyref.fq_to_mont(pyref.FQ2_NONRESIDUE)
    assert [L(v) for v in g2["WEIERSTRASS_B_MONT"]] == [pyref.fq_to_mont(v) for v in pyref.G2_B]
    g2gen = (
        (pyref.fq_from_mont(L(g2["G2_GENERATOR_X_C0_MONT"])), pyref.fq_from_mont(L(g2["G2_GENERATOR_X_C1_MONT"]))),
        (pyref.fq_from_mont(L(g2["G2_GENERATOR_Y_C0_MONT"])), pyref.fq_from_mont(L(g2["G2_GENERATOR_Y_C1_MONT"]))),
    )
    assert pyref.g2_is_on_curve(g2gen)
    raw = golden["beta_h_g2"]
    f = lambda o: int.from_bytes(raw[o : o + 48], "little")
    beta_h = ((f(0), f(48)), (f(96), f(144) & ((1 << 382) - 1)))
    assert pyref.g2_is_on_curve(beta_h)


def _bases(golden, n):
    pts = util.srs_points_ints(golden["srs_g1"], n)
    return pts, util.g1_affine_from_ints(pts)


@pytest.mark.parametrize("n", [1, 5, 10, 14, 15, 31, 32, 50, 100])
def test_msm_all_variants_agree_small(golden, n):
    """variable_base/mod.rs:91-106 (test_msm): naive == standard == batched, compared after to_affine."""
    pts, aff = _bases(golden, n)
    sc = synthetic.random_fr_integers(n, 1000 + n)
    want = pyref.msm_naive(pts, util.fr_to_ints(sc))
    for kind in (oracle.MSM_NAIVE, oracle.MSM_STANDARD, oracle.MSM_BATCHED):
        got = util.g1_affine_to_ints(oracle.g1_to_affine(oracle.g1_msm(aff, sc, kind)))[0]
        assert got == want, (n, kind)


@pytest.mark.parametrize("n", [500, 1000, 1024])
def test_msm_variants_agree_medium(golden, n):
    _, aff = _bases(golden, n)
    sc = synthetic.random_fr_integers(n, 2000 + n)
    ref = oracle.g1_to_affine(oracle.g1_msm(aff, sc, oracle.MSM_NAIVE))
    assert util.affine_equal(oracle.g1_to_affine(oracle.g1_msm(aff, sc, oracle.MSM_STANDARD)), ref)
    assert util.affine_equal(oracle.g1_to_affine(oracle.g1_msm(aff, sc, oracle.MSM_BATCHED)), ref)


def test_msm_config1_full_real_srs_differential(golden):
    """SURVEY.md 8(d) config 1 as written: the 32 768 real SRS points followed by their negations (2^16 bases) - the closest thing
    to a reference-held MSM input (the reference holds no MSM KAT, msm/tests.rs:27-67 is differential).  batched::msm ==
    standard::msm on random scalars; and with equal scalars on P_i and -P_i the sum is the point at infinity."""
    bases = util.srs_config1_bases(golden["srs_g1_full"])
    n = bases.shape[0]
    assert n == 1 << 16 and oracle.g1_is_on_curve(bases[::257]) and oracle.g1_is_on_curve(bases[-3:])
    sc = synthetic.random_fr_integers(n, 0xC0F1)
    a = oracle.g1_to_affine(oracle.g1_msm(bases, sc, oracle.MSM_BATCHED))
    b = oracle.g1_to_affine(oracle.g1_msm(bases, sc, oracle.MSM_STANDARD))
    assert util.affine_equal(a, b) and not int(a["infinity"][0])
    sc[n // 2 :] = sc[: n // 2]
    z = oracle.g1_to_affine(oracle.g1_msm(bases, sc, oracle.MSM_BATCHED))
    assert int(z["infinity"][0]) == 1


def test_msm_unequal_lengths(golden):
    """msm/tests.rs:54-67: more bases than scalars - extra bases ignored."""
    _, aff = _bases(golden, 1024)
    sc = synthetic.random_fr_integers(924, 7)
    ref = oracle.g1_to_affine(oracle.g1_msm(aff[:924], sc, oracle.MSM_NAIVE))
    assert util.affine_equal(oracle.g1_to_affine(oracle.g1_msm(aff, sc, oracle.MSM_BATCHED)), ref)
    assert util.affine_equal(oracle.g1_to_affine(oracle.g1_msm(aff, sc, oracle.MSM_STANDARD)), ref)


def test_msm_edge_cases(golden):
    """scalars 0/1/r-1, duplicate bases (P+P), P and -P, infinity bases."""
    pts, _ = _bases(golden, 64)
    r = pyref.R_MOD
    pts = pts[:40] + [pts[3]] * 8 + [pyref.g1_neg(pts[5])] * 4 + [None] * 4 + pts[40:48]
    scal = [0, 1, r - 1, 2, 1, 1, r - 1, 0] + [int(v) for v in synthetic.splitmix64(3, 56)]
    scal[40:48] = [scal[3]] * 8  # same scalar on duplicate bases -> same bucket -> doubling branch
    scal[48:52] = [scal[5]] * 4  # P and -P in one bucket
    aff = util.g1_affine_from_ints(pts)
    sc = util.ints_to_fr(scal)
    want = pyref.msm_naive(pts, [s % r for s in scal])
    for kind in (oracle.MSM_NAIVE, oracle.MSM_STANDARD, oracle.MSM_BATCHED):
        got = util.g1_affine_to_ints(oracle.g1_to_affine(oracle.g1_msm(aff, sc, kind)))[0]
        assert got == want, kind
    # all-zero scalars -> infinity, canonical affine zero (0, 1, inf)
    z = oracle.g1_to_affine(oracle.g1_msm(aff, np.zeros((64, 4), dtype=np.uint64), oracle.MSM_BATCHED))
    assert z["infinity"][0] == 1 and L(z["y"][0]) == pyref.fq_to_mont(1) and L(z["x"][0]) == 0


def test_gen_bases_and_structured_msm():
    """bases (i+1)*G: sum_i s_i (i+1) G == (sum_i s_i (i+1) mod r) G - an O(n) closed form usable at any size."""
    g = util.g1_generator_affine()
    n = 300
    bases = oracle.g1_gen_bases(g, 1, n)
    ints = util.g1_affine_to_ints(bases[:5])
    assert ints[0] == pyref.G1_GEN and ints[4] == pyref.g1_mul(pyref.G1_GEN, 5)
    sc = synthetic.random_fr_integers(n, 42)
    k = sum((i + 1) * s for i, s in enumerate(util.fr_to_ints(sc))) % pyref.R_MOD
    want = oracle.g1_to_affine(oracle.g1_mul(g, util.limbs(k, 4)))
    got = oracle.g1_to_affine(oracle.g1_msm(bases, sc, oracle.MSM_BATCHED))
    assert util.affine_equal(got, want)


def test_g2_msm_small(golden):
    g2 = golden["constants"]["g2"]
    gen = (
        (pyref.fq_from_mont(L(g2["G2_GENERATOR_X_C0_MONT"])), pyref.fq_from_mont(L(g2["G2_GENERATOR_X_C1_MONT"]))),
        (pyref.fq_from_mont(L(g2["G2_GENERATOR_Y_C0_MONT"])), pyref.fq_from_mont(L(g2["G2_GENERATOR_Y_C1_MONT"]))),
    )
    pts = [pyref.g2_mul(gen, k) for k in (1, 2, 3, 5, 7, 11, 13, 17)] + [None]
    scal = [int(v) % pyref.R_MOD for v in synthetic.splitmix64(77, 9)]
    scal[1] = 1
    scal[2] = 0
    want = pyref.msm_naive_g2(pts, scal)
    aff = util.g2_affine_from_ints(pts)
    sc = util.ints_to_fr(scal)
    for kind in (oracle.MSM_NAIVE, oracle.MSM_STANDARD):
        got = util.g2_affine_to_ints(oracle.g2_to_affine(oracle.g2_msm(aff, sc, kind)))[0]
        assert got == want
    # 40 bases -> windowed path (c = 1 for < 32 scalars is covered above; here c = ln(40)+2)
    pts40 = [pyref.g2_mul(gen, k + 1) for k in range(40)]
    sc40 = synthetic.random_fr_integers(40, 9)
    want = pyref.msm_naive_g2(pts40, util.fr_to_ints(sc40))
    got = util.g2_affine_to_ints(oracle.g2_to_affine(oracle.g2_msm(util.g2_affine_from_ints(pts40), sc40, oracle.MSM_STANDARD)))[0]
    assert got == want
