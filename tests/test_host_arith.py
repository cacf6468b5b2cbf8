"""The gfx950 device arithmetic (snarkvm_amd/csrc/ff.hip.h, ec.hip.h: 29-bit limbs) compiled for the host and
executed on the CPU through the snarkvm_hip_selftest_* hooks, compared with the oracle.  No GPU needed."""
import ctypes
import re
import os

import numpy as np
import pytest

from oracle import cpu as oracle
from oracle import pyref
from snarkvm_amd import _lib, synthetic
from tests import util

OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "inverse": 4, "neg": 5, "from_bigint": 6, "to_bigint": 7, "lazy_chain": 8, "diff_of_products": 9}


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def host_field(field, op, a, b=None):
    L = _lib.lib()
    nl = 4 if field == 0 else 6
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, nl)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, nl)
    out = np.zeros_like(a)
    rc = L.snarkvm_hip_selftest_field(ctypes.c_int(field), ctypes.c_int(OPS[op]), _p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    assert rc == 0
    return out


def _rand_mont(field, n, seed):
    mod = pyref.R_MOD if field == 0 else pyref.Q_MOD
    nl = 4 if field == 0 else 6
    rng = np.random.default_rng(seed)
    vals = [int.from_bytes(rng.bytes(48), "little") % mod for _ in range(n)]
    vals[:4] = [0, 1, mod - 1, mod - 2]
    return np.array([pyref.to_limbs(v, nl) for v in vals], dtype=np.uint64)


@pytest.mark.parametrize("field", [0, 1])
def test_device_field_arithmetic_on_host_matches_oracle(field):
    ofn = oracle.fr_op if field == 0 else oracle.fq_op
    a = _rand_mont(field, 300, 10 + field)
    b = _rand_mont(field, 300, 20 + field)[::-1].copy()
    for op in ("add", "sub", "mul"):
        assert np.array_equal(host_field(field, op, a, b), ofn(op, a, b)), op
    assert np.array_equal(host_field(field, "sqr", a), ofn("sqr", a))
    assert np.array_equal(host_field(field, "neg", a), ofn("neg", a))
    assert np.array_equal(host_field(field, "from_bigint", a), ofn("from_bigint", a))
    assert np.array_equal(host_field(field, "to_bigint", a), ofn("to_bigint", a))
    nz = a[1:40]
    assert np.array_equal(host_field(field, "inverse", nz), ofn("inverse", nz))
    # a*b - c*d with a single (signed-accumulator) Montgomery reduction, used by the XYZZ addition law
    want = ofn("sub", ofn("mul", a, b), ofn("mul", b, ofn("add", a, b)))
    assert np.array_equal(host_field(field, "diff_of_products", a, b), want)
    if field == 0:  # lazy add / sub / mul + final reduction (NTT butterflies) == plain product
        assert np.array_equal(host_field(field, "lazy_chain", a, b), ofn("mul", a, b))


def test_limb_tables_match_reference_constants(golden):
    """Every 29-bit table in ff.hip.h is re-derived from the reference's constants."""
    src = open(os.path.join(util.ROOT, "snarkvm_amd", "csrc", "ff.hip.h")).read()

    def table(struct, name):
        body = src[src.index("struct " + struct):]
        body = body[: body.index("\n};")]
        m = re.search(r"uint32_t " + name + r"\[\d+\] = \{(.*?)\}", body, re.S)
        vals = [int(x.rstrip("u"), 0) for x in re.findall(r"0x[0-9a-fA-F]+u?|\b\d+\b", m.group(1))]
        return sum(v << (29 * i) for i, v in enumerate(vals))

    for struct, key, nlimb, membits in (("FrP", "fr", 9, 256), ("FqP", "fq", 13, 384)):
        mod = pyref.from_limbs(golden["constants"][key]["MODULUS"])
        B = 29 * nlimb
        assert table(struct, "MOD") == mod
        assert table(struct, "ONE") == pow(2, B, mod)
        assert table(struct, "R2") == pow(2, 2 * B, mod)
        assert table(struct, "MEM2INT") == pow(2, 2 * B - membits, mod)
        assert table(struct, "INT2MEM") == pow(2, membits, mod)
        assert pyref.from_limbs(golden["constants"][key]["R"]) == pow(2, membits, mod)
    # two-adic root and generator constants used by ntt.hip.h / api.hip
    ntt = open(os.path.join(util.ROOT, "snarkvm_amd", "csrc", "ntt.hip.h")).read()
    m = re.search(r"FR_TWO_ADIC_ROOT_MEM\[8\] = \{(.*?)\}", ntt, re.S)
    words = [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
    assert sum(w << (32 * i) for i, w in enumerate(words)) == pyref.from_limbs(golden["constants"]["fr"]["TWO_ADIC_ROOT_OF_UNITY"])
    csrc = os.path.join(util.ROOT, "snarkvm_amd", "csrc")
    api = "".join(open(os.path.join(csrc, f)).read() for f in ("api.hip", "api_fr.hip", "api_g2.hip", "runtime.hip.h"))
    for name, key in (("G1_GEN_X", "GENERATOR_X_MONT"), ("G1_GEN_Y", "GENERATOR_Y_MONT")):
        m = re.search(name + r"\[6\] = \{(.*?)\}", api, re.S)
        assert [int(x) for x in re.findall(r"(\d+)ull", m.group(1))] == golden["constants"]["g1"][key]
    m = re.search(r"FQ_R\[6\] = \{(.*?)\}", api, re.S)
    assert [int(x) for x in re.findall(r"(\d+)ull", m.group(1))] == golden["constants"]["fq"]["R"]


def test_device_point_arithmetic_on_host(golden):
    """xyzz mixed add / add / double incl. exceptional cases, via a naive double-and-add MSM on the host."""
    L = _lib.lib()
    pts = util.srs_points_ints(golden["srs_g1"], 12)
    pts = pts + [pts[2], pyref.g1_neg(pts[4]), None, pts[0]]
    r = pyref.R_MOD
    scal = [int(v) % r for v in synthetic.splitmix64(5, len(pts))]
    scal[0] = 0
    scal[1] = 1
    scal[3] = r - 1
    scal[12] = scal[2]  # duplicate base with the same scalar: P + P
    scal[13] = scal[4]  # P and -P with the same scalar: P + (-P)
    aff = util.g1_affine_from_ints(pts)
    sc = util.ints_to_fr(scal)
    out = np.zeros(1, dtype=oracle.G1_PROJECTIVE)
    rc = L.snarkvm_hip_selftest_g1_msm_naive(_p(aff), ctypes.c_size_t(len(pts)), ctypes.c_size_t(104), _p(sc), _p(out))
    assert rc == 0
    got = util.g1_affine_to_ints(oracle.g1_to_affine(out))[0]
    assert got == pyref.msm_naive(pts, scal)
    # all-zero scalars -> Projective::zero() = (0, 1, 0)
    rc = L.snarkvm_hip_selftest_g1_msm_naive(_p(aff), ctypes.c_size_t(len(pts)), ctypes.c_size_t(104), _p(np.zeros_like(sc)), _p(out))
    assert pyref.from_limbs(out["z"][0]) == 0 and pyref.from_limbs(out["y"][0]) == pyref.FQ_MONT_R


def test_ntt_lazy_bound():
    """Lazy butterflies keep values below 2^s * r entering local stage s; 8 stages need 256 r < 2^261 (9 x 29 bits)."""
    assert 256 * pyref.R_MOD < 1 << 261
    assert (pyref.R_MOD << 8) >> (29 * 8) < 1 << 29  # top limb of 2^8 r stays a 29-bit limb


def _plan(n, window_bits=0, tables=1, table_bits=0):
    out = (ctypes.c_uint32 * 10)()
    rc = _lib.lib().snarkvm_hip_selftest_msm_plan(ctypes.c_size_t(n), ctypes.c_int(window_bits), ctypes.c_int(tables), ctypes.c_int(table_bits), out)
    assert rc == 0
    keys = ("c", "W", "J", "Wd", "nb", "nbt", "S", "S2", "L", "wide")
    return dict(zip(keys, list(out)))


def test_msm_planner_invariants():
    """Host-side MSM planner (msm.hip.h msm_make_plan): every plan covers the 254 signed-digit bits of a scalar, wide windows
    exist only as one window per table, and the bench configurations come out as documented."""
    for n in (1, 31, 1000, 4096, 1 << 16, 1 << 20, 1 << 24):
        p = _plan(n)  # unregistered bases
        assert 2 <= p["c"] <= 16 and p["J"] == 1 and p["W"] * p["c"] >= 254 and p["nb"] == 1 << (p["c"] - 1)
        for tables in (2, 4, 8, 16):
            p = _plan(n, tables=tables)
            assert p["J"] == tables and p["W"] * p["c"] == 256 // tables and p["Wd"] == p["W"] * tables and p["c"] <= 16
    for tables, bits in ((15, 17), (15, 18), (13, 20), (12, 22), (12, 23)):
        big = _plan(1 << 24, tables=tables, table_bits=bits)
        assert big["c"] == bits and big["W"] == 1 and big["wide"] == 1 and big["Wd"] == tables and big["nb"] == 1 << (bits - 1)
        small = _plan(1000, tables=tables, table_bits=bits)   # few points: a divisor of the table width instead of a wide window
        assert bits % small["c"] == 0 and small["W"] * small["c"] == bits
        if bits not in (17, 23):  # prime widths have no other divisor >= 2
            assert small["c"] <= 16 and small["wide"] == 0
        forced = _plan(1000, window_bits=bits, tables=tables, table_bits=bits)
        assert forced["c"] == bits and forced["wide"] == 1
    assert _plan(1 << 24, tables=12, table_bits=22)["S"] == 128 and _plan(1 << 16, tables=16)["S"] == 16 and _plan(1 << 14, tables=16)["S"] == 4
    # table-less window choice: c <= 11 (unfolded tail) up to 2^17 points, 16 beyond (the top window stays full)
    assert _plan(1 << 12)["c"] == 8 and _plan(1 << 16)["c"] == 11 and _plan(1 << 17)["c"] == 11 and _plan((1 << 17) + 1)["c"] == 16 and _plan(1 << 21)["c"] == 16
    # segments fill whole rounds of one wave per SIMD: no 1 088-wave grids on 1 024 SIMDs
    assert _plan(70000, tables=16)["S"] == 18 and _plan(1 << 18, tables=16)["S"] == 64 and _plan(1 << 19, tables=16)["S"] == 64 and _plan((1 << 18) + 5, tables=16)["S"] == 33
    d = _plan(1 << 20, window_bits=13, tables=16)  # request that does not divide the table width: largest divisor below
    assert d["c"] == 8 and d["W"] == 2


def test_msm_host_finish_matches_oracle():
    """The host-side end of every MSM (runtime.hip.h msm_accum_t: bit-plane sums at bit positions -> one Horner chain; also the
    combine step of a multi-device / chunked MSM, the reference's host `dadd`, snarkvm.cu:290-295) compiled for the host:
    sum_i 2^pos[i] * P_i == the oracle's MSM with scalars 2^pos[i] mod r, for repeated positions, infinity planes, position 0
    and positions beyond the scalar field's bit length."""
    from snarkvm_amd.layout import G1_PROJECTIVE

    rng = np.random.default_rng(77)
    n = 40
    bases = oracle.g1_gen_bases(util.g1_generator_affine(), 3, n)
    pos = rng.integers(0, 300, size=n).astype(np.int32)
    pos[:6] = [0, 0, 299, 299, 17, 253]
    planes = np.zeros(n, dtype=G1_PROJECTIVE)
    planes["x"] = bases["x"]
    planes["y"] = bases["y"]
    planes["z"] = np.array(pyref.to_limbs(pyref.fq_to_mont(1), 6), dtype=np.uint64)
    inf = [5, 11]
    for i in inf:  # Projective::zero(): z == 0 (any x, y)
        planes[i]["z"] = 0
    out = np.zeros(1, dtype=G1_PROJECTIVE)
    rc = _lib.lib().snarkvm_hip_selftest_g1_finish(_p(planes), _p(pos), ctypes.c_size_t(n), _p(out))
    assert rc == 0
    keep = [i for i in range(n) if i not in inf]
    scalars = util.ints_to_fr([pow(2, int(pos[i]), pyref.R_MOD) for i in keep])
    want = oracle.g1_msm(bases[keep], scalars)
    assert util.affine_equal(oracle.g1_to_affine(out), oracle.g1_to_affine(want))
    # nothing but infinity -> Projective::zero()
    rc = _lib.lib().snarkvm_hip_selftest_g1_finish(_p(planes[inf]), _p(pos[inf].copy()), ctypes.c_size_t(2), _p(out))
    assert rc == 0 and oracle.g1_to_affine(out)["infinity"][0] == 1


def test_lazy_accumulate_arithmetic_matches_exact():
    """csrc/ffl.hip.h (signed 29-bit limbs, R = 2^406, no canonical form inside the accumulate loop) against the exact
    arithmetic of ff.hip.h / ec.hip.h, compiled for the host: chains of mixed additions with doublings, cancellations and
    restarts from infinity, every coordinate compared after every step; then the field routines at the edges of their ranges."""
    L = _lib.lib()
    for seed in (1, 2, 3, 0xDEADBEEF):
        assert L.snarkvm_hip_selftest_fq_lazy(ctypes.c_uint64(seed), ctypes.c_int(5000)) == 0, seed


def test_lazy_tail_arithmetic_matches_exact():
    """csrc/ffl.hip.h::fqz_t (the lazy arithmetic behind field-like operators: the reduce rounds, fold and bit planes of a G1 MSM under
    tuning lazy_tail) against the exact arithmetic under the same generic addition and doubling laws, compiled for the host: general
    representatives, repeated points, negatives, restarts from infinity, explicit doublings, the 208-byte memory image after every step."""
    L = _lib.lib()
    for seed in (1, 2, 3, 0xDEADBEEF):
        assert L.snarkvm_hip_selftest_g1_lazy_tail(ctypes.c_uint64(seed), ctypes.c_int(4000)) == 0, seed


def test_signed_ntt_butterfly_arithmetic_matches_exact():
    """csrc/frs.hip.h (signed limbs, R = 2^290, no canonical form inside a pass) compiled for the host against the exact Fr
    arithmetic (which test_field_ops pins on the oracle): chains of up to nine butterfly stages, the closing product, the bare
    reduction and the folded closing-table form, including operands with all-ones / all-zero limbs."""
    L = _lib.lib()
    for seed in (1, 0xFEED, 0x5EED5EED, 2**63 + 11):
        assert L.snarkvm_hip_selftest_fr_signed(ctypes.c_uint64(seed), ctypes.c_int(400)) == 0, seed


def test_lazy_g2_accumulate_arithmetic_matches_exact():
    """csrc/ffl2.hip.h (lazy Fq2 on signed limbs: the factor 5 of the non-residue folded into a 14-limb operand, two-product
    reductions, tight-range subtractions) compiled for the host against the exact Fq2 / XYZZ arithmetic: chains of G2 mixed additions
    with doublings, cancellations and restarts, every coordinate after every step; products, squares and the raw partial-sum image."""
    L = _lib.lib()
    pts = synthetic.g2_points(48, distinct=48)
    for seed in (1, 7, 0xC0FFEE):
        rc = L.snarkvm_hip_selftest_fq2_lazy(ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(pts.shape[0]), ctypes.c_uint64(seed), ctypes.c_int(1500))
        assert rc == 0, (seed, rc)


def test_fq2_lane_pair_arithmetic_matches_exact():
    """csrc/ffl2p.hip.h (round 5): the G2 accumulate kernel's Fq2 arithmetic split over a lane pair - c0 on the even lane, c1 on the odd
    lane, one two-product column sum per lane, operands exchanged between the lanes - run on the host with both lanes side by side
    (the same source, exchange = a swap of two array entries) against the exact Fq2 / XYZZ arithmetic: chains of G2 mixed additions with
    doublings and cancellations (resolved inside the pair arithmetic, mdbl-2008-s-1) and restarts, every component after every step."""
    L = _lib.lib()
    pts = synthetic.g2_points(48, distinct=48)
    for seed in (1, 7, 0xC0FFEE, 99):
        rc = L.snarkvm_hip_selftest_fq2_pair(ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(pts.shape[0]), ctypes.c_uint64(seed), ctypes.c_int(1500))
        assert rc == 0, (seed, rc)


def test_g2_sixteen_lane_cooperative_addition_matches_exact():
    """csrc/hex2.hip.h (round 6): the Fq2 XYZZ addition of the G2 tail trees dealt to the SIXTEEN lanes of a DPP row - lane 4 q + p computes Fq sub-product p
    of the quad schedule's product q, a pair exchange, c0 = P0 - 5 P1 / c1 = P2 + P3, a gather per round - run on the host over sixteen simulated lanes
    (the same source, the exchanges index the other lanes' copies) against xyzz_t<fq2_t>::add: every coordinate on every lane, operands at infinity,
    P + P and P - P (every lane sees the equal x coordinates and abandons the addition) included; and the flagged plain addition of the tail kernels
    (xyzz_t::add_flag): equal x coordinates raise the flag and leave the accumulator alone, everything else is the exact sum."""
    L = _lib.lib()
    pts = synthetic.g2_points(48, distinct=48)
    for seed in (1, 7, 0xC0FFEE, 2026):
        rc = L.snarkvm_hip_selftest_g2_hex(ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(pts.shape[0]), ctypes.c_uint64(seed), ctypes.c_int(600))
        assert rc == 0, (seed, rc)
