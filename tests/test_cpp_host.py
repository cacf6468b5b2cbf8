"""include/snarkvm_hip.hpp - the C++ host mirror of the Rust plugin crate - built with plain g++ against the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from tests import util

SRC = os.path.join(util.ROOT, "tests", "cpp", "hpp_host.cpp")
LIBDIR = os.path.join(util.ROOT, "snarkvm_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "hpp_host")
    rocm = "/opt/rocm/lib"
    cmd = ["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(util.ROOT, "include"), SRC, "-o", exe, "-L", LIBDIR, "-lsnarkvm_hip",
           f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{rocm}", f"-Wl,-rpath-link,{rocm}"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_cpp_mirror_validates_arguments_and_fails_loudly(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "validate"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_matches_oracle_on_device(tmp_path, golden):
    from oracle import cpu as oracle
    from snarkvm_amd import synthetic

    exe = _build(tmp_path)
    n = 3000
    bases = oracle.g1_gen_bases(util.g1_generator_affine(), 5, n)
    sc = synthetic.random_fr_integers(n, 1212)
    fr = oracle.fr_op("from_bigint", synthetic.random_fr_integers(1 << 10, 1313))
    bases.tofile(tmp_path / "bases.bin")
    sc.tofile(tmp_path / "scalars.bin")
    fr.tofile(tmp_path / "fr.bin")
    r = subprocess.run([exe, "run", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(tmp_path / "msm.bin", dtype=oracle.G1_PROJECTIVE)
    assert util.affine_equal(oracle.g1_to_affine(got), oracle.g1_to_affine(oracle.g1_msm(bases, sc)))
    assert np.array_equal(np.fromfile(tmp_path / "ntt.bin", dtype=np.uint64).reshape(-1, 4), oracle.ntt(fr))
    want = oracle.polymul(10, [fr[:512], fr[512:]])
    assert np.array_equal(np.fromfile(tmp_path / "polymul.bin", dtype=np.uint64).reshape(-1, 4), want)
    # device-resident operands owned through snarkvm_hip::DeviceBuffer inside a Scope (no HIP header, no torch in that process)
    padded = np.concatenate([fr, np.zeros_like(fr)])
    assert np.array_equal(np.fromfile(tmp_path / "ntt_dev.bin", dtype=np.uint64).reshape(-1, 4), oracle.ntt(padded))
    got_dev = np.fromfile(tmp_path / "msm_dev.bin", dtype=oracle.G1_PROJECTIVE)
    assert util.affine_equal(oracle.g1_to_affine(got_dev), oracle.g1_to_affine(oracle.g1_msm(bases, sc)))


@pytest.mark.gpu
def test_cpp_two_logical_devices_four_caller_threads(tmp_path, golden):
    """The C ABI alone (no Python in the loop) with two logical devices and four concurrent caller threads: point-range split of
    `snarkvm_msm` over both devices, base-cache registration on both, lanes handed to concurrent callers - every result
    bit-exact vs the oracle."""
    from oracle import cpu as oracle
    from snarkvm_amd import synthetic

    exe = _build(tmp_path)
    n = (1 << 19) + 1234  # >= 2 * MSM_SPLIT_MIN: the call is cut into two point-range chunks, one per device
    bases = oracle.g1_gen_bases(util.g1_generator_affine(), 1, n)
    sc = synthetic.random_fr_integers(n, 4242)
    fr = oracle.fr_op("from_bigint", synthetic.random_fr_integers(1 << 12, 4343))
    bases.tofile(tmp_path / "bases.bin")
    sc.tofile(tmp_path / "scalars.bin")
    fr.tofile(tmp_path / "fr.bin")
    r = subprocess.run([exe, "multi", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    for t in range(4):
        rot = np.roll(sc, -7 * t, axis=0)
        k = util.weighted_sum_mod_r(rot, start=1)  # bases are (i + 1) G
        want = oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), util.limbs(k, 4)))
        for rep in range(3):
            got = np.fromfile(tmp_path / f"msm_{t}_{rep}.bin", dtype=oracle.G1_PROJECTIVE)
            assert util.affine_equal(oracle.g1_to_affine(got), want), (t, rep)
        x = np.roll(fr, -3 * t, axis=0)
        assert np.array_equal(np.fromfile(tmp_path / f"ntt_{t}.bin", dtype=np.uint64).reshape(-1, 4), oracle.ntt(x)), t
