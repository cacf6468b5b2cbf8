"""Device (de)serialisation of G1 points (snarkvm_amd/csrc/serde.hip.h through the C ABI) against the Python oracle and
the real SRS bytes, on an MI355X."""
import numpy as np
import pytest

from oracle import cpu as oracle
from oracle import pyref
from snarkvm_amd import serialize, synthetic
from snarkvm_amd.msm import RegisteredBases
from tests import util

pytestmark = pytest.mark.gpu


def test_usrs_bytes_decode_on_device(golden):
    raw = bytes(golden["srs_g1"])
    n = len(raw) // 96
    got = serialize.g1_deserialize(raw, compressed=False)
    want = util.g1_affine_from_ints(util.srs_points_ints(raw, n))
    assert util.affine_equal(got, want)
    assert util.g1_affine_to_ints(got[:1])[0] == pyref.G1_GEN
    # the reference's `Valid::check` (on curve + prime-order subgroup) holds for real SRS points
    assert util.affine_equal(serialize.g1_deserialize(raw[: 96 * 64], compressed=False, validate=True), want[:64])
    # and encoding is the inverse
    assert serialize.g1_serialize(got, compressed=False) == raw


def test_compressed_roundtrip_matches_oracle(golden):
    pts = util.srs_points_ints(golden["srs_g1"], 200)
    pts = pts + [pyref.g1_neg(p) for p in pts[:50]] + [None, None]
    aff = util.g1_affine_from_ints(pts)
    comp = serialize.g1_serialize(aff, compressed=True)
    assert comp == b"".join(pyref.g1_serialize(p, True) for p in pts)
    back = serialize.g1_deserialize(comp, compressed=True)
    assert util.affine_equal(back, aff)
    unc = serialize.g1_serialize(aff, compressed=False)
    assert unc == b"".join(pyref.g1_serialize(p, False) for p in pts)
    assert util.affine_equal(serialize.g1_deserialize(unc, compressed=False), aff)


def test_decode_errors_on_device():
    gen = pyref.g1_serialize(pyref.G1_GEN, compressed=True)
    bad = bytearray(gen)
    bad[47] |= 0xC0
    with pytest.raises(serialize.SerializationError):
        serialize.g1_deserialize(gen + bytes(bad), compressed=True)
    with pytest.raises(serialize.SerializationError):
        serialize.g1_deserialize(pyref.Q_MOD.to_bytes(48, "little"), compressed=True)
    x = 1
    while pyref.fq_sqrt((x ** 3 + 1) % pyref.Q_MOD) is not None:
        x += 1
    with pytest.raises(serialize.SerializationError):
        serialize.g1_deserialize(x.to_bytes(48, "little"), compressed=True)
    x = 2
    while True:
        y = pyref.fq_sqrt((x ** 3 + 1) % pyref.Q_MOD)
        if y is not None and pyref.g1_mul((x, y), pyref.R_MOD) is not None:
            break
        x += 1
    enc = pyref.g1_serialize((x, y), compressed=False)
    assert util.g1_affine_to_ints(serialize.g1_deserialize(enc, compressed=False)) == [(x, y)]
    with pytest.raises(serialize.SerializationError):
        serialize.g1_deserialize(enc, compressed=False, validate=True)
    with pytest.raises(serialize.SerializationError):  # off-curve
        serialize.g1_deserialize((5).to_bytes(48, "little") + (7).to_bytes(48, "little"), compressed=False, validate=True)


@pytest.mark.parametrize("compressed", [False, True])
def test_msm_over_bases_registered_from_bytes(golden, compressed):
    """SRS bytes -> device base slots -> MSM, equal to the oracle's MSM over the decoded points."""
    raw = bytes(golden["srs_g1"])
    n = len(raw) // 96
    pts = util.srs_points_ints(raw, n)
    aff = util.g1_affine_from_ints(pts)
    data = b"".join(pyref.g1_serialize(p, True) for p in pts) if compressed else raw
    scalars = synthetic.random_fr_integers(n, 606)
    want = oracle.g1_to_affine(oracle.g1_msm(aff, scalars))
    for tables in (1, 16):
        rb = RegisteredBases.from_serialized(data, n, compressed=compressed, tables=tables)
        assert util.affine_equal(oracle.g1_to_affine(rb.msm(scalars)), want)
        rb.close()
    count, body = serialize.split_usrs(len(pts).to_bytes(8, "little") + raw)
    assert count == n and bytes(body) == raw


def test_g2_uncompressed_on_device(golden):
    """`beta-h.usrs` and synthetic G2 points through the device decoder / encoder, against the Python oracle."""
    from tests.test_gpu_parity import _g2_bases

    raw = bytes(golden["beta_h_g2"])
    got = serialize.g2_deserialize(raw, validate=True)
    assert util.g2_affine_to_ints(got) == [pyref.g2_deserialize(raw)]
    assert serialize.g2_serialize(got) == raw
    pts = _g2_bases(golden, 40)
    pts[3]["infinity"] = 1
    ints = util.g2_affine_to_ints(pts)
    enc = serialize.g2_serialize(pts)
    # the encoder writes the coordinates of an infinity record as they are; the oracle writes (0, 1): compare the others
    for i, p in enumerate(ints):
        if p is not None:
            assert enc[192 * i : 192 * i + 192] == pyref.g2_serialize(p)
        else:
            assert enc[192 * i + 191] >> 6 == 1
    back = serialize.g2_deserialize(enc)
    assert util.g2_affine_to_ints(back) == ints
    bad = bytearray(enc[:192])
    bad[191] |= 0xC0
    with pytest.raises(serialize.SerializationError):
        serialize.g2_deserialize(bytes(bad))
    off = bytearray(enc[:192])
    off[0] ^= 1  # no longer on the curve
    assert len(serialize.g2_deserialize(bytes(off))) == 1
    with pytest.raises(serialize.SerializationError):
        serialize.g2_deserialize(bytes(off), validate=True)


def test_g2_compressed_on_device(golden):
    """Compressed G2 points (96 B) through the device decoder / encoder: `Fp2::sqrt` + the sign rule of fp2.rs:240-250 on the
    device against the Python oracle - the real `beta-h.usrs` point, generator multiples, their negatives (both sign flags),
    infinity, and every decode error (both flag bits, x >= q, an x with no point on the curve, a point outside the subgroup)."""
    from tests.test_gpu_parity import _g2_bases

    raw = bytes(golden["beta_h_g2"])
    beta_h = serialize.g2_deserialize(raw, validate=True)
    pts = np.concatenate([beta_h, _g2_bases(golden, 30)])
    neg = pts.copy()
    ints = util.g2_affine_to_ints(pts)
    q = pyref.Q_MOD
    neg_ints = [(x, ((-y[0]) % q, (-y[1]) % q)) for x, y in ints]
    neg = util.g2_affine_from_ints(neg_ints)
    both = np.concatenate([pts, neg])
    both_ints = ints + neg_ints
    enc = serialize.g2_serialize(both, compressed=True)
    assert len(enc) == 96 * len(both_ints)
    for i, p in enumerate(both_ints):
        assert enc[96 * i : 96 * i + 96] == pyref.g2_serialize_compressed(p), i
    assert {enc[96 * i + 95] >> 7 for i in range(len(both_ints))} == {0, 1}
    back = serialize.g2_deserialize(enc, validate=True, compressed=True)
    assert util.g2_affine_to_ints(back) == both_ints
    # infinity
    inf = serialize.g2_deserialize(pyref.g2_serialize_compressed(None), compressed=True)
    assert inf[0]["infinity"] == 1 and util.g2_affine_to_ints(inf) == [None]
    both[2]["infinity"] = 1
    e2 = serialize.g2_serialize(both[2:3], compressed=True)
    assert e2 == pyref.g2_serialize_compressed(None)
    # errors
    bad = bytearray(enc[:96])
    bad[95] |= 0xC0
    with pytest.raises(serialize.SerializationError):
        serialize.g2_deserialize(bytes(bad), compressed=True)
    big = bytearray(enc[:96])
    big[:48] = (q + 5).to_bytes(48, "little")
    with pytest.raises(serialize.SerializationError):
        serialize.g2_deserialize(bytes(big), compressed=True)
    # an x coordinate whose x^3 + b' has no square root (found with the oracle)
    x = (3, 1)
    while pyref.fq2_sqrt(pyref.fq2_add(pyref.fq2_mul(pyref.fq2_mul(x, x), x), pyref.G2_B)) is not None:
        x = (x[0] + 1, 1)
    with pytest.raises(serialize.SerializationError):
        serialize.g2_deserialize(x[0].to_bytes(48, "little") + x[1].to_bytes(48, "little"), compressed=True)
    # on the curve but outside the prime-order subgroup: accepted unchecked, rejected with validation
    x = (7, 2)
    while True:
        y = pyref.fq2_sqrt(pyref.fq2_add(pyref.fq2_mul(pyref.fq2_mul(x, x), x), pyref.G2_B))
        if y is not None and pyref.g2_mul((x, y), pyref.R_MOD) is not None:
            break
        x = (x[0] + 1, 2)
    off = pyref.g2_serialize_compressed((x, y))
    got = serialize.g2_deserialize(off, compressed=True)
    assert util.g2_affine_to_ints(got) == [pyref.g2_deserialize_compressed(off)]
    with pytest.raises(serialize.SerializationError):
        serialize.g2_deserialize(off, compressed=True, validate=True)
