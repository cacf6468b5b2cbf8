"""`SonicKZG10::{commit, batch_open}` mirror (snarkvm_amd/sonic_pc.py) against the reference's formulas evaluated with the
oracle: every commitment is `msm(powers view[lz..], to_bigint(coeffs[lz..])) + msm(gamma view, to_bigint(blinding))`
(kzg10/mod.rs:98-156 through sonic_pc/mod.rs:177-257, degree-bounded polynomials over the shifted powers starting at
`max_bound - bound`, data_structures.rs:310-331); an opening proof commits to the witness of the challenge-weighted
combination (sonic_pc/mod.rs:259-342, kzg10/mod.rs:213-322)."""
import numpy as np
import pytest

from oracle import cpu as oracle
from snarkvm_amd import kzg10, sonic_pc, synthetic
from tests import util

pytestmark = pytest.mark.gpu


def _rnd(n, seed):
    return oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, seed))


def _linear_divisor(point):
    one = oracle.fr_op("from_bigint", np.array([[1, 0, 0, 0]], dtype=np.uint64))
    return [(0, oracle.fr_op("neg", np.asarray(point).reshape(1, 4))[0]), (1, one[0])]


class Challenges:
    """Stand-in for the Fiat-Shamir sponge: hands out a fixed list of Fr challenges in order."""

    def __init__(self, seed, n=64):
        self.vals = _rnd(n, seed)
        self.vals[1] = sonic_pc.FR_ONE  # `coeff.is_one()` takes the plain-addition branch (mod.rs:555-557)
        self.k = 0

    def squeeze_short_nonnative_field_element(self):
        v = self.vals[self.k]
        self.k += 1
        return v


def _expect_commit(bases, lz_coeffs, gamma, blind):
    nz = np.nonzero(lz_coeffs.any(axis=1))[0]
    acc = None
    if nz.size:
        lz = int(nz[0])
        acc = oracle.g1_msm(bases[lz : lz + lz_coeffs.shape[0] - lz], oracle.fr_op("to_bigint", lz_coeffs[lz:]))
    if blind is not None and blind.shape[0]:
        h = oracle.g1_msm(gamma[: blind.shape[0]], oracle.fr_op("to_bigint", blind))
        acc = h if acc is None else oracle.g1_add(acc, h)
    if acc is None:
        out = np.zeros(1, dtype=oracle.G1_AFFINE)
        out["y"] = util.g1_affine_from_ints([None])["y"]
        out["infinity"] = 1
        return out
    return oracle.g1_to_affine(acc)


def test_sonic_commit_and_batch_open_match_reference_formulas():
    G = util.g1_generator_affine()
    N = 700  # max_degree + 1
    powers = oracle.g1_gen_bases(G, 1, N)
    gamma = oracle.g1_gen_bases(G, 5000, 6)
    bounds = [200, 450, 699]
    shifted = oracle.g1_gen_bases(G, 9000, bounds[-1] + 1)
    shifted_gamma = {b: oracle.g1_gen_bases(G, 20000 + b, 4) for b in bounds}
    lag = {64: oracle.g1_gen_bases(G, 30000, 64)}
    ck = sonic_pc.CommitterUnionKey(powers, gamma, shifted, shifted_gamma, bounds, lag)
    blinds = iter([_rnd(4, 11), _rnd(3, 12), _rnd(5, 13)])  # hiding_bound + 2 coefficients each (degree hiding_bound + 1)
    rng = lambda k: next(blinds)[:k]  # noqa: E731
    p_plain = _rnd(N, 1)
    p_lead = _rnd(300, 2)
    p_lead[:17] = 0                       # leading zeros are skipped together with their bases (kzg10/mod.rs:455-467)
    p_bound = _rnd(180, 3)                # degree 179 <= bound 200, hiding
    p_maxb = _rnd(bounds[-1] + 1, 4)      # degree == the largest bound
    p_zero = np.zeros((0, 4), dtype=np.uint64)
    evals = _rnd(64, 5)
    polys = [
        sonic_pc.LabeledPolynomial("plain", p_plain),
        sonic_pc.LabeledPolynomial("hiding", p_lead, hiding_bound=2),
        sonic_pc.LabeledPolynomial("bounded", p_bound, degree_bound=200, hiding_bound=1),
        sonic_pc.LabeledPolynomial("maxbound", p_maxb, degree_bound=bounds[-1]),
        sonic_pc.LabeledPolynomial("zero", p_zero),
        sonic_pc.LabeledEvaluations("lagrange", evals, hiding_bound=3),
    ]
    comms, rands = sonic_pc.SonicKZG10.commit(N - 1, ck, polys, rng)
    assert [c.label for c in comms] == [p.label for p in polys] and [c.degree_bound for c in comms] == [None, None, 200, bounds[-1], None, None]
    b = [r.blinding_polynomial for r in rands]
    assert b[0].shape[0] == 0 and b[1].shape[0] == 4 and b[2].shape[0] == 3 and b[5].shape[0] == 5
    start = bounds[-1] - 200
    want = [
        _expect_commit(powers, p_plain, gamma, None),
        _expect_commit(powers, p_lead, gamma, b[1]),
        _expect_commit(shifted[start:], p_bound, shifted_gamma[200], b[2]),
        _expect_commit(shifted, p_maxb, shifted_gamma[bounds[-1]], None),
        _expect_commit(powers, p_zero, gamma, None),
        _expect_commit(lag[64], evals, gamma, b[5]),
    ]
    for c, w, p in zip(comms, want, polys):
        assert util.affine_equal(np.array([c.commitment]), w), p.label

    # ---- error behaviour of check_degrees_and_bounds / the size checks (kzg10/mod.rs:407-452)
    with pytest.raises(kzg10.PCError):
        sonic_pc.SonicKZG10.commit(N - 1, ck, [sonic_pc.LabeledPolynomial("bad", _rnd(10, 9), degree_bound=123)])        # bound not enforced
    with pytest.raises(kzg10.PCError):
        sonic_pc.SonicKZG10.commit(N - 1, ck, [sonic_pc.LabeledPolynomial("bad", _rnd(300, 9), degree_bound=200)])       # degree > bound
    with pytest.raises(kzg10.PCError):
        sonic_pc.SonicKZG10.commit(N - 1, ck, [sonic_pc.LabeledPolynomial("bad", _rnd(N + 1, 9))])                       # too many coefficients
    with pytest.raises(kzg10.PCError):
        sonic_pc.SonicKZG10.commit(N - 1, ck, [sonic_pc.LabeledPolynomial("bad", _rnd(10, 9), hiding_bound=1)])          # hiding without rng
    with pytest.raises(kzg10.PCError):
        sonic_pc.SonicKZG10.commit(N - 1, ck, [sonic_pc.LabeledEvaluations("bad", _rnd(32, 9))])                         # no Lagrange basis of that size

    # ---- batch_open: two query points; "hiding" is opened at both
    open_polys = polys[:3]
    open_rands = rands[:3]
    z1, z2 = _rnd(1, 21), _rnd(1, 22)
    query_set = [("plain", ("beta", z1)), ("hiding", ("beta", z1)), ("hiding", ("alpha", z2)), ("bounded", ("alpha", z2))]
    proofs = sonic_pc.SonicKZG10.batch_open(N - 1, ck, open_polys, query_set, open_rands, Challenges(99))
    chal = Challenges(99)
    by_label = {p.label: (p, r) for p, r in zip(open_polys, open_rands)}
    for proof, (name, point, labels) in zip(proofs, [("alpha", z2, ["bounded", "hiding"]), ("beta", z1, ["hiding", "plain"])]):
        length = max(by_label[lb][0].coeffs.shape[0] for lb in labels)
        comb = np.zeros((length, 4), dtype=np.uint64)
        rlen = max(by_label[lb][1].blinding_polynomial.shape[0] for lb in labels) or 1
        comb_r = np.zeros((rlen, 4), dtype=np.uint64)
        for lb in labels:
            c = chal.squeeze_short_nonnative_field_element()
            pc = np.zeros((length, 4), dtype=np.uint64)
            pc[: by_label[lb][0].coeffs.shape[0]] = by_label[lb][0].coeffs
            comb = oracle.fr_vec_op("axpy", comb, pc, np.tile(c, (length, 1)))
            rb = np.zeros((rlen, 4), dtype=np.uint64)
            bl = by_label[lb][1].blinding_polynomial
            rb[: bl.shape[0]] = bl
            comb_r = oracle.fr_vec_op("axpy", comb_r, rb, np.tile(c, (rlen, 1)))
        chal.squeeze_short_nonnative_field_element()  # the unused `_randomizer`
        wq, _ = oracle.poly_divide(comb, _linear_divisor(point))
        want_w = oracle.g1_msm(powers[: wq.shape[0]], oracle.fr_op("to_bigint", wq))
        bq, _ = oracle.poly_divide(comb_r, _linear_divisor(point))
        if bq.shape[0]:
            want_w = oracle.g1_add(want_w, oracle.g1_msm(gamma[: bq.shape[0]], oracle.fr_op("to_bigint", bq)))
        assert util.affine_equal(np.array([proof.w]), oracle.g1_to_affine(want_w)), name
        assert np.array_equal(proof.random_v, oracle.poly_evaluate(comb_r, point)), name
    with pytest.raises(kzg10.PCError):
        sonic_pc.SonicKZG10.batch_open(N - 1, ck, open_polys, [("nope", ("beta", z1))], open_rands, Challenges(1))
    ck.close()
