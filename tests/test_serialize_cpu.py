"""Canonical G1 encoding: the Python oracle (oracle/pyref.py) against the real SRS bytes committed under tests/golden,
and the constants of snarkvm_amd/csrc/serde.hip.h re-derived from the reference's field parameters."""
import os
import re

import pytest

from oracle import pyref
from tests import util


def _words(src, name):
    m = re.search(name + r"\[\d+\] = \{(.*?)\}", src, re.S)
    vals = [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-fA-F]+u?", m.group(1))]
    return sum(v << (32 * i) for i, v in enumerate(vals))


def test_serde_constants_match_reference(golden):
    src = open(os.path.join(util.ROOT, "snarkvm_amd", "csrc", "serde.hip.h")).read()
    fq = golden["constants"]["fq"]
    q = pyref.from_limbs(fq["MODULUS"])
    t = pyref.from_limbs(fq["T"])
    assert q - 1 == t << fq["TWO_ADICITY"] and fq["TWO_ADICITY"] == 46
    assert _words(src, "FQ_T_MINUS_ONE_DIV_TWO") == (t - 1) // 2
    root = pyref.fq_from_mont(pyref.from_limbs(fq["TWO_ADIC_ROOT_OF_UNITY"]))
    assert _words(src, "FQ_TWO_ADIC_ROOT_INT") == root
    assert pow(root, 1 << 46, q) == 1 and pow(root, 1 << 45, q) == q - 1
    assert _words(src, "FR_MODULUS_WORDS") == pyref.from_limbs(golden["constants"]["fr"]["MODULUS"])


def test_real_srs_bytes_decode_to_curve_points(golden):
    """powers-of-beta-15.usrs: point 0 is the G1 generator, every decoded point is on the curve, re-encoding is the identity."""
    raw = golden["srs_g1"]
    pts = [pyref.g1_deserialize(raw[96 * i : 96 * i + 96], compressed=False) for i in range(64)]
    assert pts[0] == pyref.G1_GEN
    assert all(pyref.g1_is_on_curve(p) for p in pts)
    assert pts == util.srs_points_ints(raw, 64)
    for i, p in enumerate(pts):
        assert pyref.g1_serialize(p, compressed=False) == bytes(raw[96 * i : 96 * i + 96])
    assert pyref.g1_deserialize(raw[:96], compressed=False, validate=True) == pyref.G1_GEN


def test_compressed_roundtrip_and_sign_rule(golden):
    pts = util.srs_points_ints(golden["srs_g1"], 16)
    for p in pts + [pyref.g1_neg(p) for p in pts[:4]] + [None]:
        c = pyref.g1_serialize(p, compressed=True)
        assert len(c) == 48
        assert pyref.g1_deserialize(c, compressed=True) == p
        if p is not None:
            assert bool(c[47] >> 7) == (p[1] > pyref.Q_MOD - p[1])  # bit 7 = y is the larger root
    u = pyref.g1_serialize(None, compressed=False)
    assert u[95] == 1 << 6 and pyref.g1_deserialize(u, compressed=False) is None


def test_decode_errors():
    bad = bytearray(pyref.g1_serialize(pyref.G1_GEN, compressed=True))
    bad[47] |= 0xC0
    with pytest.raises(pyref.SerializationError):
        pyref.g1_deserialize(bytes(bad), compressed=True)
    with pytest.raises(pyref.SerializationError):  # x >= q
        pyref.g1_deserialize((pyref.Q_MOD).to_bytes(48, "little"), compressed=True)
    # an x with no point above it
    x = 1
    while pyref.fq_sqrt((x ** 3 + 1) % pyref.Q_MOD) is not None:
        x += 1
    with pytest.raises(pyref.SerializationError):
        pyref.g1_deserialize(x.to_bytes(48, "little"), compressed=True)
    # a curve point outside the prime-order subgroup fails validation only
    x = 2
    while True:
        y = pyref.fq_sqrt((x ** 3 + 1) % pyref.Q_MOD)
        if y is not None and pyref.g1_mul((x, y), pyref.R_MOD) is not None:
            break
        x += 1
    enc = pyref.g1_serialize((x, y), compressed=False)
    assert pyref.g1_deserialize(enc, compressed=False) == (x, y)
    with pytest.raises(pyref.SerializationError):
        pyref.g1_deserialize(enc, compressed=False, validate=True)


def test_g2_beta_h_bytes_decode(golden):
    """`beta-h.usrs` (one uncompressed G2 point): on the curve, in the prime-order subgroup, re-encoding is the identity."""
    raw = bytes(golden["beta_h_g2"])
    p = pyref.g2_deserialize(raw, validate=True)
    assert p is not None and pyref.g2_is_on_curve(p)
    assert pyref.g2_serialize(p) == raw
    assert pyref.g2_deserialize(pyref.g2_serialize(None)) is None


def test_fq2_sqrt_and_compressed_g2_oracle(golden):
    """The Python oracle of the compressed G2 encoding, pinned on reference-held data: `Fp2::sqrt` (fp2.rs:208-230) returns a
    root of every square and None for non-squares; compress -> decompress is the identity on the real `beta-h.usrs` point, on
    the G2 generator's multiples and on their negatives (both values of the sign flag)."""
    import random

    rnd = random.Random(5)
    q = pyref.Q_MOD
    for _ in range(12):
        a = (rnd.randrange(q), rnd.randrange(q))
        sq = pyref.fq2_mul(a, a)
        r = pyref.fq2_sqrt(sq)
        assert r is not None and pyref.fq2_mul(r, r) == sq
    assert pyref.fq2_sqrt((4, 0)) in ((2, 0), (q - 2, 0))
    nonsquares = 0
    for _ in range(24):
        a = (rnd.randrange(q), rnd.randrange(1, q))
        r = pyref.fq2_sqrt(a)
        if r is None:
            nonsquares += 1
        else:
            assert pyref.fq2_mul(r, r) == a
    assert 4 <= nonsquares <= 20  # about half of Fq2 are non-squares
    p = pyref.g2_deserialize(bytes(golden["beta_h_g2"]), validate=True)
    pts = [p, pyref.g2_mul(p, 2), pyref.g2_mul(p, 12345)]
    pts += [(x, ((-y[0]) % q, (-y[1]) % q)) for x, y in pts]
    flags = set()
    for pt in pts:
        enc = pyref.g2_serialize_compressed(pt)
        assert len(enc) == 96
        flags.add(enc[95] >> 7)
        assert pyref.g2_deserialize_compressed(enc, validate=True) == pt
    assert flags == {0, 1}
    assert pyref.g2_deserialize_compressed(pyref.g2_serialize_compressed(None)) is None
    bad = bytearray(pyref.g2_serialize_compressed(p))
    bad[95] |= 0xC0
    with pytest.raises(pyref.SerializationError):
        pyref.g2_deserialize_compressed(bytes(bad))
