"""Setup-time group operations on the device (snarkvm_amd/csrc/group.hip.h through the C ABI) against the oracle:
FixedBase::msm and the group-element iFFT behind UniversalParams::lagrange_basis."""
import numpy as np
import pytest

from oracle import cpu as oracle
from oracle import pyref
from snarkvm_amd import fft, group, kzg10, synthetic
from tests import util
from tests.test_gpu_parity import _srs

pytestmark = pytest.mark.gpu


def _rnd(n, seed):
    return oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, seed))


@pytest.mark.parametrize("n", [1, 31, 1000, 20000])
def test_fixed_base_msm(n):
    """FixedBase::msm (msm/fixed_base.rs:87-97): [v_i * g] for the reference's own window rule."""
    g = util.g1_generator_affine()
    v = _rnd(n, 300 + n)
    v[0] = 0
    if n > 3:
        v[1] = util.ints_to_fr_mont([1])[0]
        v[2] = util.ints_to_fr_mont([pyref.R_MOD - 1])[0]
    window = group.FixedBase.get_mul_window_size(n)
    table = group.FixedBase.get_window_table(253, window, g)
    got = group.FixedBase.msm(253, window, table, v)
    want = oracle.g1_fixed_base_msm(g, v)
    assert util.affine_equal(oracle.g1_to_affine(got), oracle.g1_to_affine(want))
    # an arbitrary (non-generator) base and the point at infinity
    base = oracle.g1_to_affine(oracle.g1_mul(g, util.limbs(123456789, 4)))
    got = group.FixedBase.msm(253, window, group.FixedBase.get_window_table(253, window, base), v[:50])
    assert util.affine_equal(oracle.g1_to_affine(got), oracle.g1_to_affine(oracle.g1_fixed_base_msm(base, v[:50])))
    inf = base.copy()
    inf["infinity"] = 1
    got = group.FixedBase.msm(253, window, group.FixedBase.get_window_table(253, window, inf), v[:5])
    assert oracle.g1_to_affine(got)["infinity"].all()


def _to_projective(aff):
    proj = np.zeros(aff.shape[0], dtype=oracle.G1_PROJECTIVE)
    proj["x"], proj["y"] = aff["x"], aff["y"]
    proj["z"] = np.array(pyref.to_limbs(pyref.fq_to_mont(1), 6), dtype=np.uint64)
    return proj


@pytest.mark.parametrize("lg", [0, 1, 2, 5, 8])
def test_group_ntt_vs_oracle(golden, lg):
    n = 1 << lg
    aff = _srs(golden, n)
    proj = _to_projective(aff)
    for inverse in (True, False):
        got = group.group_ntt(proj, inverse=inverse)
        want = oracle.g1_group_ntt(proj, inverse=inverse)
        assert util.affine_equal(oracle.g1_to_affine(got), oracle.g1_to_affine(want)), (lg, inverse)
    back = group.group_ntt(group.group_ntt(proj, inverse=True), inverse=False)
    assert util.affine_equal(oracle.g1_to_affine(back), aff)


def test_lagrange_basis_commits_like_the_monomial_basis(golden):
    """UniversalParams::lagrange_basis (kzg10/data_structures.rs:68-72): with L = iFFT(powers), committing to the
    evaluations of p over the domain with L equals committing to the coefficients of p with the powers
    (KZG10::commit_lagrange vs KZG10::commit, kzg10/mod.rs:98-206) - for any powers, since both are the same linear map."""
    lg = 10
    n = 1 << lg
    powers = _srs(golden, n)
    basis = group.lagrange_basis(powers)
    assert util.affine_equal(basis, oracle.g1_to_affine(oracle.g1_group_ntt(_to_projective(powers), inverse=True)))
    coeffs = _rnd(n, 9090)
    evals = fft.EvaluationDomain.new(n).fft(coeffs)
    gamma = oracle.g1_gen_bases(util.g1_generator_affine(), 5, 2)
    pw, lb = kzg10.Powers(powers, gamma), kzg10.Powers(basis, gamma)
    c1, _ = kzg10.KZG10.commit(pw, coeffs)
    c2, _ = kzg10.KZG10.commit_lagrange(lb, evals)
    assert util.affine_equal(kzg10.to_affine(c1), kzg10.to_affine(c2))
    pw.close()
    lb.close()
