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


@pytest.mark.parametrize("lg", [12, 16])
def test_group_ntt_at_the_sizes_it_exists_for(lg):
    """UniversalParams::lagrange_basis runs at 2^16 and beyond (kzg10/data_structures.rs:68-72).  The four-lanes-per-butterfly stages (group.hip.h: quad-cooperative
    additions / doublings, 2-bit windowed twiddle multiplication) against the oracle's group iFFT and FFT at 2^12 and 2^16 - every point - plus the round trip, and
    against the one-lane-per-butterfly form of round 5 at 2^12 (tuning group_quad=0 in a child process: the same group elements)."""
    import os
    import subprocess
    import sys
    import time

    n = 1 << lg
    aff = oracle.g1_gen_bases(util.g1_generator_affine(), 3, n)
    proj = _to_projective(aff)
    t0 = time.perf_counter()
    got = group.group_ntt(proj, inverse=True)
    dt = time.perf_counter() - t0
    want = oracle.g1_group_ntt(proj, inverse=True)
    assert util.affine_equal(oracle.g1_to_affine(got), oracle.g1_to_affine(want)), lg
    fwd = group.group_ntt(got, inverse=False)
    assert util.affine_equal(oracle.g1_to_affine(fwd), aff), lg
    print(f"group iFFT 2^{lg}: {dt * 1e3:.1f} ms through the C ABI (host buffers)")
    if lg == 12:
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from snarkvm_amd import group; from snarkvm_amd.layout import G1_PROJECTIVE; "
                "p = np.fromfile(sys.argv[1], dtype=G1_PROJECTIVE); group.group_ntt(p, inverse=True).tofile(sys.argv[2])" % util.ROOT)
        src, dst = f"/tmp/group_ntt_in_{os.getpid()}.bin", f"/tmp/group_ntt_out_{os.getpid()}.bin"
        proj.tofile(src)
        r = subprocess.run([sys.executable, "-c", code, src, dst], env=dict(os.environ, SNARKVM_HIP_TUNING="group_quad=0"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        old = np.fromfile(dst, dtype=oracle.G1_PROJECTIVE)
        assert util.affine_equal(oracle.g1_to_affine(old), oracle.g1_to_affine(want))
        os.remove(src)
        os.remove(dst)
