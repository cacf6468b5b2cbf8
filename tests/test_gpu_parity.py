"""Parity of the HIP path (through the C ABI) against the oracle, on a real MI355X.  Bit-exact: NTT outputs are
compared limb for limb (unique Montgomery representation), MSM outputs after affine normalisation (x, y, infinity),
exactly like the reference's own GPU-vs-CPU tests (fft/domain.rs:1140-1218, msm/variable_base/mod.rs:109-119)."""
import ctypes
import os

import numpy as np
import pytest

from oracle import cpu as oracle
from oracle import pyref
from snarkvm_amd import _lib, batch, fft, plugin, synthetic
from snarkvm_amd.layout import G1_AFFINE, G1_PROJECTIVE, NTTDirection, NTTInputOutputOrder, NTTType
from snarkvm_amd.msm import RegisteredBases, VariableBase
from tests import util
from tests.test_host_arith import OPS, _rand_mont

pytestmark = pytest.mark.gpu


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


# ------------------------------------------------------------------------------------------ field arithmetic
@pytest.mark.parametrize("field", [0, 1])
def test_field_ops_on_device(field):
    L = _lib.lib()
    ofn = oracle.fr_op if field == 0 else oracle.fq_op
    a = _rand_mont(field, 1000, 30 + field)
    b = _rand_mont(field, 1000, 40 + field)[::-1].copy()
    for op in ("add", "sub", "mul", "sqr", "neg", "from_bigint", "to_bigint", "inverse"):
        aa = a[1:200] if op == "inverse" else a
        bb = b[1:200] if op == "inverse" else b
        out = np.zeros_like(aa)
        _lib.check(L.snarkvm_hip_devtest_field(ctypes.c_int(field), ctypes.c_int(OPS[op]), _p(aa), _p(bb), _p(out), ctypes.c_size_t(aa.shape[0])))
        want = ofn(op, aa, bb) if op in ("add", "sub", "mul") else ofn(op, aa)
        assert np.array_equal(out, want), (field, op)


# ------------------------------------------------------------------------------------------ NTT
def _fr_vec(n, seed):
    return oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, seed))


@pytest.mark.parametrize("lg", list(range(0, 20)))
def test_ntt_nn_all_transforms_vs_oracle(lg):
    """test_fft_correctness_cuda (domain.rs:1140-1218): lg 2..19, forward / inverse / coset, NN order."""
    n = 1 << lg
    x = _fr_vec(n, 100 + lg)
    for d in (NTTDirection.Forward, NTTDirection.Inverse):
        for t in (NTTType.Standard, NTTType.Coset):
            got = x.copy()
            plugin.NTT(n, got, NTTInputOutputOrder.NN, d, t)
            assert np.array_equal(got, oracle.ntt(x, oracle.ORDER_NN, d, t)), (lg, d, t)


@pytest.mark.parametrize("lg", [3, 8, 9, 13, 17])
def test_ntt_all_orders(lg):
    n = 1 << lg
    x = _fr_vec(n, 200 + lg)
    for order in (NTTInputOutputOrder.NR, NTTInputOutputOrder.RN, NTTInputOutputOrder.RR):
        for d in (NTTDirection.Forward, NTTDirection.Inverse):
            got = x.copy()
            plugin.NTT(n, got, order, d, NTTType.Standard)
            assert np.array_equal(got, oracle.ntt(x, order, d, oracle.STANDARD)), (lg, order, d)


@pytest.mark.parametrize("lg", [0, 5, 12, 16, 18])
def test_ntt_device_batch_vs_oracle(lg):
    """snarkvm_hip_ntt_device_batch: several device vectors, mixed directions and types, one enqueue and one synchronisation -
    every vector equals the oracle's transform; a vector listed twice is transformed twice in list order (round trip)."""
    import torch

    n = 1 << lg
    xs = [_fr_vec(n, 7000 + 10 * lg + i) for i in range(5)]
    dev = [torch.from_numpy(x.view(np.int64).copy()).cuda() for x in xs]
    torch.cuda.synchronize()
    dirs = [NTTDirection.Forward, NTTDirection.Inverse, NTTDirection.Forward, NTTDirection.Inverse, NTTDirection.Forward]
    tys = [NTTType.Standard, NTTType.Standard, NTTType.Coset, NTTType.Coset, NTTType.Standard]
    plugin.NTT_device_batch(lg, [t.data_ptr() for t in dev], dirs, tys)
    for i in range(5):
        got = dev[i].cpu().numpy().view(np.uint64).reshape(-1, 4)
        assert np.array_equal(got, oracle.ntt(xs[i], oracle.ORDER_NN, dirs[i], tys[i])), (lg, i)
    # the same vector forward then inverse in one batch: back to the forward input
    before = dev[0].cpu().numpy().copy()
    plugin.NTT_device_batch(lg, [dev[0].data_ptr(), dev[0].data_ptr()], [NTTDirection.Inverse, NTTDirection.Forward], None)
    assert np.array_equal(dev[0].cpu().numpy(), before)
    plugin.NTT_device_batch(lg, [])  # empty batch: nothing to do
    # defaults (NULL directions / types) = forward, standard
    plugin.NTT_device_batch(lg, [dev[1].data_ptr()])
    want = oracle.ntt(oracle.ntt(xs[1], oracle.ORDER_NN, NTTDirection.Inverse, NTTType.Standard), oracle.ORDER_NN, NTTDirection.Forward, NTTType.Standard)
    assert np.array_equal(dev[1].cpu().numpy().view(np.uint64).reshape(-1, 4), want)


@pytest.mark.parametrize("lg", [20, 21, 22, 23])
def test_ntt_large_vs_oracle(lg):
    """BASELINE configs[2] (domain sizes 2^18 - 2^24): every size of the range that the lg 0..19 sweep above and the 2^24 test
    below do not cover - pass splits 7+7+6, 7+7+7, 8+7+7, 8+8+7 - every element of all four transforms against the oracle,
    like the reference's own sweep (fft/domain.rs:1140-1218), plus the round trips."""
    n = 1 << lg
    x = _fr_vec(n, 300 + lg)
    for d in (NTTDirection.Forward, NTTDirection.Inverse):
        for t in (NTTType.Standard, NTTType.Coset):
            got = x.copy()
            plugin.NTT(n, got, NTTInputOutputOrder.NN, d, t)
            assert np.array_equal(got, oracle.ntt(x, oracle.ORDER_NN, d, t)), (lg, d, t)
            plugin.NTT(n, got, NTTInputOutputOrder.NN, NTTDirection.Inverse if d == NTTDirection.Forward else NTTDirection.Forward, t)
            assert np.array_equal(got, x), (lg, d, t, "round trip")


def test_ntt_2_24_properties():
    """BASELINE size: round trip, coset round trip and one spot value by Horner (size-independent properties)."""
    lg = 24
    n = 1 << lg
    x = _fr_vec(n, 324)
    y = x.copy()
    plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Forward, NTTType.Standard)
    # X[0] = sum of inputs, X[n/2] = alternating sum
    ints = None
    s = oracle.fr_op("add", x[0::2], x[1::2])
    while s.shape[0] > 1:
        s = oracle.fr_op("add", s[0::2], s[1::2])
    assert np.array_equal(y[0], s[0])
    z = y.copy()
    plugin.NTT(n, z, NTTInputOutputOrder.NN, NTTDirection.Inverse, NTTType.Standard)
    assert np.array_equal(z, x)
    plugin.NTT(n, z, NTTInputOutputOrder.NN, NTTDirection.Forward, NTTType.Coset)
    plugin.NTT(n, z, NTTInputOutputOrder.NN, NTTDirection.Inverse, NTTType.Coset)
    assert np.array_equal(z, x)


def test_ntt_2_24_and_2_25_vs_oracle():
    """The BASELINE size (three radix-2^8 passes) and the first size served by radix-2^9 passes, every element against the
    oracle's `fft_in_place` / `ifft_in_place` restatement (all four transforms at 2^24, forward + coset inverse at 2^25)."""
    for lg, kinds in ((24, [(NTTDirection.Forward, NTTType.Standard), (NTTDirection.Inverse, NTTType.Standard), (NTTDirection.Forward, NTTType.Coset),
                            (NTTDirection.Inverse, NTTType.Coset)]),
                      (25, [(NTTDirection.Forward, NTTType.Standard), (NTTDirection.Inverse, NTTType.Coset)])):
        n = 1 << lg
        x = _fr_vec(n, 4000 + lg)
        for direction, kind in kinds:
            y = x.copy()
            plugin.NTT(n, y, NTTInputOutputOrder.NN, direction, kind)
            assert np.array_equal(y, oracle.ntt(x, oracle.ORDER_NN, direction, kind)), (lg, direction, kind)


def test_ntt_2_26_round_trip():
    """The largest supported domain (2 GiB): forward + inverse and coset forward + coset inverse return the input."""
    n = 1 << 26
    x = _fr_vec(n, 4026)
    y = x.copy()
    plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Forward, NTTType.Standard)
    assert not np.array_equal(y[:1024], x[:1024])
    plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Inverse, NTTType.Standard)
    assert np.array_equal(y, x)
    plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Forward, NTTType.Coset)
    plugin.NTT(n, y, NTTInputOutputOrder.NN, NTTDirection.Inverse, NTTType.Coset)
    assert np.array_equal(y, x)


def test_ntt_rejects_oversized_domain():
    with pytest.raises(_lib.HipError):
        _lib.check(_lib.lib().snarkvm_ntt(None, ctypes.c_uint32(27), 0, 0, 0))


def test_kat_intt8_and_domain_wrappers(golden):
    evals = [1, 2, 8, 4, 32, 2, 128, 0]
    z_lde = [int(v) for v in golden["varuna"]["polynomials"]["z_lde"]]
    dom = fft.EvaluationDomain.new(8)
    got = dom.ifft(util.ints_to_fr_mont(evals))
    assert util.fr_mont_to_ints(got) == z_lde
    # resize semantics: truncation and zero padding (domain.rs:171)
    short = util.ints_to_fr_mont([5, 7, 11])
    assert np.array_equal(dom.fft(short), oracle.ntt(np.vstack([short, np.zeros((5, 4), dtype=np.uint64)])))
    assert fft.EvaluationDomain.new(0).size == 1 and fft.EvaluationDomain.new(1 << 48) is None


def test_polymul_vs_oracle(golden):
    rng = np.random.default_rng(7)
    for lens in [(3, 5), (17, 40), (64, 64), (1000, 900, 100), (5000, 3000)]:
        polys = [_fr_vec(k, 400 + k) for k in lens]
        lg = 0
        while (1 << lg) < sum(lens):
            lg += 1
        got = plugin.polymul(1 << lg, polys, [])
        assert np.array_equal(got, oracle.polymul(lg, polys)), lens
    # coefficient + evaluation form, and the single-evaluation corner case (snarkvm.cu:203-208)
    polys = [_fr_vec(100, 1), _fr_vec(120, 2)]
    ev = [_fr_vec(256, 3)]
    assert np.array_equal(plugin.polymul(256, polys, ev), oracle.polymul(8, polys, ev))
    assert np.array_equal(plugin.polymul(256, [], ev), oracle.polymul(8, [], ev))
    # KAT-polymul16 through PolyMultiplier (trim of trailing zeros)
    pa = oracle.ntt(util.ints_to_fr_mont([2, 2, 2, 2, 2, 8, 32, 0]), direction=oracle.INVERSE)
    pb = oracle.ntt(util.ints_to_fr_mont([4, 4, 4, 4, 4, 4, 4, 0]), direction=oracle.INVERSE)
    pm = fft.PolyMultiplier()
    pm.add_polynomial(pa)
    pm.add_polynomial(pb)
    res = pm.multiply()
    want = oracle.polymul(4, [pa, pb])
    assert np.array_equal(res, want[: res.shape[0]]) and not want[res.shape[0] :].any()


# ------------------------------------------------------------------------------------------ MSM
def _srs(golden, n):
    pts = util.srs_points_ints(golden["srs_g1"])
    aff = util.g1_affine_from_ints(pts)
    reps = (n + len(pts) - 1) // len(pts)
    return np.tile(aff, reps)[:n].copy()  # tiled like the reference benches (benches/msm/variable_base.rs:29-32)


def _check(bases, scalars, got):
    want = oracle.g1_to_affine(oracle.g1_msm(bases, scalars, oracle.MSM_BATCHED))
    assert util.affine_equal(oracle.g1_to_affine(got), want)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 10, 14, 15, 31, 32, 50, 100, 500, 1000, 1024, 1025, 4096])
def test_msm_small_sizes_vs_oracle(golden, n):
    bases = _srs(golden, n)
    sc = synthetic.random_fr_integers(n, 500 + n)
    _check(bases, sc, VariableBase.msm(bases, sc))


@pytest.mark.parametrize("lg", [14, 16, 18])
def test_msm_pow2_vs_oracle(golden, lg):
    """test_msm_cuda (variable_base/mod.rs:109-119) sizes and the 2^16 plumbing config of BASELINE.json."""
    n = 1 << lg
    g = util.g1_generator_affine()
    bases = oracle.g1_gen_bases(g, 1, n)
    sc = synthetic.random_fr_integers(n, synthetic.SEED_MSM_2_16)
    _check(bases, sc, VariableBase.msm(bases, sc))


def test_msm_unequal_lengths(golden):
    bases = _srs(golden, 1024)
    sc = synthetic.random_fr_integers(924, 9)
    got = VariableBase.msm(bases, sc)
    _check(bases[:924], sc, got)


def test_msm_edge_cases(golden):
    pts = util.srs_points_ints(golden["srs_g1"], 64)
    r = pyref.R_MOD
    pts = pts[:40] + [pts[3]] * 8 + [pyref.g1_neg(pts[5])] * 4 + [None] * 4 + pts[40:48]
    scal = [0, 1, r - 1, 2, 1, 1, r - 1, 0] + [int(v) % r for v in synthetic.splitmix64(3, 56)]
    scal[40:48] = [scal[3]] * 8
    scal[48:52] = [scal[5]] * 4
    bases = util.g1_affine_from_ints(pts)
    sc = util.ints_to_fr(scal)
    got = util.g1_affine_to_ints(oracle.g1_to_affine(VariableBase.msm(bases, sc)))[0]
    assert got == pyref.msm_naive(pts, scal)
    # all-zero scalars and the empty MSM -> infinity, affine (0, 1, inf)
    for b_, s_ in ((bases, np.zeros((64, 4), dtype=np.uint64)), (bases, np.zeros((0, 4), dtype=np.uint64))):
        z = oracle.g1_to_affine(VariableBase.msm(b_, s_))
        assert z["infinity"][0] == 1
    # all scalars equal (one bucket per window takes every point) and all scalars == r - 1
    n = 3000
    bases = _srs(golden, n)
    for v in (12345678901234567890123, r - 1, 1, 2**252):
        sc = util.ints_to_fr([v] * n)
        _check(bases, sc, VariableBase.msm(bases, sc))


def test_msm_witness_like_distribution(golden):
    n = 1 << 15
    bases = oracle.g1_gen_bases(util.g1_generator_affine(), 7, n)
    sc = synthetic.witness_like_scalars(n, 77)
    _check(bases, sc, VariableBase.msm(bases, sc))


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 16])
def test_msm_every_window_size(golden, c):
    n = 2000
    bases = _srs(golden, n)
    sc = synthetic.random_fr_integers(n, 600 + c)
    rb = RegisteredBases(bases)
    _check(bases, sc, rb.msm(sc, window_bits=c))
    # sub-range of the registered vector (KZG degree-bounded commitments use a base offset, kzg10/mod.rs:124-129)
    _check(bases[100:1100], sc[:1000], rb.msm(sc[:1000], offset=100, window_bits=c))
    rb.close()


@pytest.mark.parametrize("tables", [2, 4, 8, 16])
def test_msm_precomputed_tables(golden, tables):
    """Registered bases with 2^(256/tables * j) multiples: same group element, shorter serial tail."""
    n = 5000
    bases = _srs(golden, n)
    bases[17]["infinity"] = 1
    sc = synthetic.random_fr_integers(n, 700 + tables)
    sc[:3] = util.ints_to_fr([0, 1, pyref.R_MOD - 1])
    rb = RegisteredBases(bases, tables=tables)
    _check(bases, sc, rb.msm(sc))
    _check(bases[40:140], sc[:100], rb.msm(sc[:100], offset=40))      # small n -> c = 8
    _check(bases, sc, rb.msm(sc, window_bits=4))
    rb.close()


def test_msm_batch_pipelined(golden):
    """A batch of independent MSMs (ragged sizes, offsets) pipelined over several streams == one-at-a-time results."""
    bases = _srs(golden, 6000)
    rb = RegisteredBases(bases, tables=4)
    sizes = [6000, 1, 333, 4096, 0, 5000, 17, 2048]
    offs = [0, 5, 100, 1000, 0, 1000, 3000, 7]
    scal = [synthetic.random_fr_integers(k, 1300 + i) for i, k in enumerate(sizes)]
    got = rb.msm_batch(scal, offsets=offs)
    for i, (k, o) in enumerate(zip(sizes, offs)):
        want = oracle.g1_to_affine(oracle.g1_msm(bases[o:o + k], scal[i], oracle.MSM_BATCHED)) if k else None
        a = oracle.g1_to_affine(got[i:i + 1])
        if k == 0:
            assert a["infinity"][0] == 1
        else:
            assert util.affine_equal(a, want), i
    rb.close()


@pytest.mark.parametrize("tables,window_bits", [(17, 15), (16, 0), (20, 13)])
def test_msm_batch_fused_multi_instance(golden, tables, window_bits):
    """Batches of proof-sized instances over windowed tables run FUSED (runtime.hip.h::msm_batch_run: one launch sequence per
    group, instance id = top sort key).  Ragged sizes (0, 1, tile boundaries 8191 / 8192 / 8193, > 2^16), base offsets, host and
    device scalars, Montgomery scalars, and an instance too big to fuse in the middle of the batch; every result against the
    oracle's batched::msm of that instance alone."""
    import torch

    N = 70000
    bases = _srs(golden, N)
    bases[33]["infinity"] = 1
    rb = RegisteredBases(bases, tables=tables, window_bits=window_bits)
    sizes = [8192, 1, 0, 8191, 8193, 70000, 333, 65536, 40000, 2, 16384]
    offs = [0, 5, 0, 100, 1000, 0, 60000, 17, 30000, 69998, 8192]
    scal = [synthetic.random_fr_integers(k, 2300 + i) for i, k in enumerate(sizes)]
    scal[0][:3] = util.ints_to_fr([0, 1, pyref.R_MOD - 1])
    scal[3][:] = scal[3][0]          # all scalars equal: one bucket per table takes every point of the instance
    want = [oracle.g1_to_affine(oracle.g1_msm(bases[o:o + k], s, oracle.MSM_BATCHED)) if k else None for k, o, s in zip(sizes, offs, scal)]

    def check(got, label):
        for i, k in enumerate(sizes):
            a = oracle.g1_to_affine(got[i:i + 1])
            if k == 0:
                assert a["infinity"][0] == 1, (label, i)
            else:
                assert util.affine_equal(a, want[i]), (label, i, k)

    check(rb.msm_batch(scal, offsets=offs), "host scalars")
    d_sc = [torch.from_numpy(s.view(np.int64).copy()).cuda() if s.shape[0] else torch.zeros(4, dtype=torch.int64, device="cuda") for s in scal]
    torch.cuda.synchronize()
    check(rb.msm_batch(device_ptrs=[d.data_ptr() for d in d_sc], npoints=sizes, offsets=offs), "device scalars")
    mont = [oracle.fr_op("from_bigint", s) if s.shape[0] else s for s in scal]
    check(rb.msm_batch(mont, offsets=offs, montgomery=True), "montgomery scalars")
    # the same instances one at a time through the single-MSM path
    for i, (k, o) in enumerate(zip(sizes, offs)):
        if k:
            assert util.affine_equal(oracle.g1_to_affine(rb.msm(scal[i], offset=o)), want[i]), ("single", i)
    # a group bigger than one fused launch takes (40 instances > MSM_FUSE_MAX_K)
    many = [scal[6]] * 40
    got = rb.msm_batch(many, offsets=[offs[6]] * 40)
    for i in range(40):
        assert util.affine_equal(oracle.g1_to_affine(got[i:i + 1]), want[6]), ("many", i)
    rb.close()


def _device_bases(n, start=1):
    import torch

    buf = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _lib.check(_lib.lib().snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(start), ctypes.c_size_t(n)))
    return buf


def test_generated_bases_match_oracle():
    n = 5000
    buf = _device_bases(n, start=3)
    got = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=G1_AFFINE)
    want = oracle.g1_gen_bases(util.g1_generator_affine(), 3, n)
    assert util.affine_equal(got, want)


@pytest.mark.parametrize("lg", [20, 24])
def test_msm_full_size_closed_form(lg):
    """BASELINE sizes.  bases_i = (i+1) G, so sum_i s_i bases_i = (sum_i s_i (i+1) mod r) G: an O(n) closed form
    that needs no CPU MSM.  Scalars and bases are device-resident (registered)."""
    import torch

    n = 1 << lg
    buf = _device_bases(n, start=1)
    rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=4 if lg == 24 else 1)
    sc = synthetic.random_fr_integers(n, synthetic.SEED_MSM_LARGE)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    got = rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
    k = util.weighted_sum_mod_r(sc, start=1)
    want = oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), util.limbs(k, 4)))
    assert util.affine_equal(oracle.g1_to_affine(got), want)
    if lg == 20:
        # the same vectors through the plain FFI (host pointers, bases converted per call) and the CPU oracle
        host_bases = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=G1_AFFINE)
        assert util.affine_equal(oracle.g1_to_affine(VariableBase.msm(host_bases, sc)), want)
        assert util.affine_equal(oracle.g1_to_affine(oracle.g1_msm(host_bases, sc, oracle.MSM_BATCHED)), want)
    rb.close()


@pytest.mark.parametrize("tables,window_bits", [(15, 17), (15, 18), (13, 20), (12, 22), (12, 23)])
def test_msm_wide_windows_vs_oracle(golden, tables, window_bits):
    """Registered bases with 2^(window_bits * j) tables and ONE window wider than 16 bits: u32 digits, three-level sort,
    two-axis bucket fold.  Forced (window_bits passed to the call) and automatic window choice, edge scalars included."""
    n = 40000
    bases = _srs(golden, n)
    bases[17]["infinity"] = 1
    sc = synthetic.random_fr_integers(n, 7000 + window_bits)
    r = pyref.R_MOD
    sc[0] = 0
    sc[1] = util.limbs(1, 4)
    sc[2] = util.limbs(r - 1, 4)
    sc[3] = util.limbs((1 << 252) + 12345, 4)
    sc[4] = sc[5]
    sc[100:200] = 0
    rb = RegisteredBases(bases, tables=tables, window_bits=window_bits)
    want = oracle.g1_to_affine(oracle.g1_msm(bases, sc))
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc, window_bits=window_bits)), want)   # wide path forced
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc)), want)                            # planner's choice
    # sub-range with an offset, tiny and empty inputs through the same handle
    want2 = oracle.g1_to_affine(oracle.g1_msm(bases[1000:1777], sc[:777]))
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc[:777], offset=1000, window_bits=window_bits)), want2)
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc[:777], offset=1000)), want2)
    one = oracle.g1_to_affine(oracle.g1_msm(bases[5:6], sc[5:6]))
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc[5:6], offset=5, window_bits=window_bits)), one)
    assert oracle.g1_to_affine(rb.msm(sc[:0]))["infinity"][0] == 1
    # batch API over the wide tables
    got = rb.msm_batch([sc, sc[:12345]], window_bits=window_bits)
    assert util.affine_equal(oracle.g1_to_affine(got[0:1]), want)
    assert util.affine_equal(oracle.g1_to_affine(got[1:2]), oracle.g1_to_affine(oracle.g1_msm(bases[:12345], sc[:12345])))
    rb.close()


def test_msm_wide_windows_skewed_buckets(golden):
    """All scalars equal / tiny: every digit lands in a handful of buckets (multi-round partial reduction on the wide path)."""
    n = 30000
    bases = _srs(golden, n)
    rb = RegisteredBases(bases, tables=12, window_bits=22)
    for val in (3, (1 << 21) + 1, pyref.R_MOD - 2):
        sc = np.tile(util.limbs(val, 4), (n, 1))
        want = oracle.g1_to_affine(oracle.g1_msm(bases, sc))
        assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc, window_bits=22)), want), val
    rb.close()


def test_msm_multi_round_skewed_no_readback():
    """A multi-round MSM (> 2^22 digit entries) sizes nothing by a host read-back: two fixed reduce rounds, then the flattened-list
    fold takes whatever is left.  All scalars equal / two values / 90 % tiny at 2^19 pairs over 12 x 22-bit tables: single
    buckets with 2^19 entries; closed form over bases (i + 1) G."""
    import torch

    n = 1 << 19
    buf = _device_bases(n, start=1)
    rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=12, window_bits=22)
    del buf
    uni = synthetic.random_fr_integers(n, 9191)
    rng = np.random.default_rng(5)
    tiny = np.zeros_like(uni)
    tiny[:, 0] = rng.integers(0, 4, n, dtype=np.uint64)
    keep = rng.random(n) < 0.1
    tiny[keep] = uni[keep]
    for name, sc in (("all equal", np.tile(uni[:1], (n, 1))), ("two values", uni[rng.integers(0, 2, n)]), ("90 % tiny", tiny), ("r - 1", np.tile(util.limbs(pyref.R_MOD - 1, 4), (n, 1)))):
        d_sc = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
        torch.cuda.synchronize()
        got = rb.msm(device_ptr=d_sc.data_ptr(), npoints=n, window_bits=22)
        kk = util.weighted_sum_mod_r(sc, start=1)
        assert util.affine_equal(oracle.g1_to_affine(got), oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), util.limbs(kk, 4)))), name
    rb.close()


def test_msm_2_24_wide_closed_form():
    """2^24 pairs over 12 tables of 22-bit windows (the bench configuration), closed form as above."""
    import torch

    n = 1 << 24
    buf = _device_bases(n, start=1)
    rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=12, window_bits=22)
    del buf
    sc = synthetic.random_fr_integers(n, synthetic.SEED_MSM_LARGE)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    got = rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
    k = util.weighted_sum_mod_r(sc, start=1)
    want = oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), util.limbs(k, 4)))
    assert util.affine_equal(oracle.g1_to_affine(got), want)
    # witness-like scalars (half zero, a quarter small): heavily skewed low buckets
    sc2 = synthetic.witness_like_scalars(n, 31337)
    d_sc2 = torch.from_numpy(sc2.view(np.int64)).cuda()
    torch.cuda.synchronize()
    got2 = rb.msm(device_ptr=d_sc2.data_ptr(), npoints=n)
    k2 = util.weighted_sum_mod_r(sc2, start=1)
    assert util.affine_equal(oracle.g1_to_affine(got2), oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), util.limbs(k2, 4))))
    rb.close()


def _pseudo_random_bases(n, seed):
    """P_i = k_i G with k_i uniform in [0, r) (SplitMix64 stream): unstructured bases like the reference's differential test
    (msm/variable_base/mod.rs:109-119 samples random points), built on the device by FixedBase::msm (group.hip.h) and
    normalised by the device's batch to_affine.  Returns (Rust-layout affine bases, k as Fr Montgomery limbs)."""
    from snarkvm_amd import group, kzg10

    k = oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, seed))
    tab = group.FixedBase.get_window_table(253, 8, util.g1_generator_affine())
    return kzg10.to_affine(group.FixedBase.msm(253, 8, tab, k)), k


def _inner_product_mod_r(a_mont, b_mont):
    """sum_i a_i b_i in Fr for two (n, 4) Montgomery vectors -> canonical (1, 4) limbs (oracle field ops, tree sum)."""
    p = oracle.fr_op("mul", np.ascontiguousarray(a_mont), np.ascontiguousarray(b_mont))
    while p.shape[0] > 1:
        if p.shape[0] & 1:
            p = np.vstack([p, np.zeros((1, 4), dtype=np.uint64)])
        p = oracle.fr_op("add", np.ascontiguousarray(p[0::2]), np.ascontiguousarray(p[1::2]))
    return oracle.fr_op("to_bigint", p)


def test_msm_2_20_pseudo_random_bases():
    """BASELINE configs[1] size 2^20 on UNSTRUCTURED bases P_i = k_i G: sum_i s_i P_i = (sum_i s_i k_i mod r) G.  Removes the
    dependence of every large-size check on the one base family (i + 1) G.  Table-less FFI path (host buffers, chunked),
    registered 16 x 16-bit tables, and the oracle's batched::msm on a 2^16 slice of the same vectors."""
    n = 1 << 20
    bases, k = _pseudo_random_bases(n, 0xBA5E5)
    # sanity of the construction itself against the oracle (first points)
    first = oracle.g1_to_affine(np.concatenate([oracle.g1_mul(util.g1_generator_affine(), oracle.fr_op("to_bigint", k[i:i + 1])[0]) for i in range(4)]))
    assert util.affine_equal(bases[:4], first)
    sc = synthetic.random_fr_integers(n, 0x5CA1A)
    want = oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), _inner_product_mod_r(oracle.fr_op("from_bigint", sc), k)[0]))
    assert util.affine_equal(oracle.g1_to_affine(VariableBase.msm(bases, sc)), want)
    rb = RegisteredBases(bases, tables=16)
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc)), want)
    m = 1 << 16
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc[:m])), oracle.g1_to_affine(oracle.g1_msm(bases[:m], sc[:m], oracle.MSM_BATCHED)))
    rb.close()


def test_msm_config1_real_srs_points_and_negations(golden):
    """SURVEY.md 8(d) config 1 exactly as written: the 32 768 points of powers-of-beta-15.usrs followed by their negations
    (2^16 bases, benches/msm/variable_base.rs:29-32), random scalars.  The table-less FFI symbol, registered 17 x 15-bit tables as
    a single call, and the same MSM as one instance of a fused batch (beside a shorter instance over the same points), all
    against the oracle's batched::msm; and P_i, -P_i with equal scalars sum to the point at infinity on every path."""
    bases = util.srs_config1_bases(golden["srs_g1_full"])
    n = bases.shape[0]
    sc = synthetic.random_fr_integers(n, 0xC0F1)
    want = oracle.g1_to_affine(oracle.g1_msm(bases, sc, oracle.MSM_BATCHED))
    assert util.affine_equal(oracle.g1_to_affine(VariableBase.msm(bases, sc)), want)
    rb = RegisteredBases(bases, tables=17, window_bits=15)
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc)), want)
    m = 40000
    res = rb.msm_batch([sc, sc[:m], sc], offsets=[0, 100, 0])
    assert util.affine_equal(oracle.g1_to_affine(res[0:1]), want) and util.affine_equal(oracle.g1_to_affine(res[2:3]), want)
    assert util.affine_equal(oracle.g1_to_affine(res[1:2]), oracle.g1_to_affine(oracle.g1_msm(bases[100 : 100 + m], sc[:m], oracle.MSM_BATCHED)))
    sc2 = sc.copy()
    sc2[n // 2 :] = sc2[: n // 2]
    for got in (VariableBase.msm(bases, sc2), rb.msm(sc2), rb.msm_batch([sc2, sc2])[1:2]):
        assert int(oracle.g1_to_affine(got)["infinity"][0]) == 1
    rb.close()


def test_msm_2_24_pseudo_random_bases():
    """BASELINE configs[1] at its full size on UNSTRUCTURED bases: P_i = k_i G with k_i from a SplitMix64 stream (built on the
    device in four slabs), 12 x 22-bit tables, uniform scalars: sum_i s_i P_i = (sum_i s_i k_i mod r) G.  Every other 2^24 check uses
    the structured family (i + 1) G."""
    import torch

    n, slab = 1 << 24, 1 << 22
    dev = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    ks = []
    for j in range(n // slab):
        b, k = _pseudo_random_bases(slab, 0xBA5E7 + j)
        dev[j * slab * G1_AFFINE.itemsize : (j + 1) * slab * G1_AFFINE.itemsize] = torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
        ks.append(k)
        del b
    torch.cuda.synchronize()
    rb = RegisteredBases(device_ptr=dev.data_ptr(), npoints=n, tables=12, window_bits=22)
    del dev
    sc = synthetic.random_fr_integers(n, 0x5CA1C)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    got = rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
    k = np.concatenate(ks)
    want = oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), _inner_product_mod_r(oracle.fr_op("from_bigint", sc), k)[0]))
    assert util.affine_equal(oracle.g1_to_affine(got), want)
    rb.close()


def test_msm_2_22_own_plan_closed_form():
    """2^22 pairs over the plan that size gets (13 tables x 20-bit windows, three sort levels, S = 64, reduce rounds) - the
    geometry was only exercised at n = 40 000 before.  Uniform and witness-like scalars, closed form over bases (i + 1) G; and
    pseudo-random bases through the same geometry at 2^21."""
    import torch

    n = 1 << 22
    buf = _device_bases(n, start=1)
    rb = RegisteredBases(device_ptr=buf.data_ptr(), npoints=n, tables=13, window_bits=20)
    del buf
    for sc in (synthetic.random_fr_integers(n, synthetic.SEED_MSM_LARGE + 22), synthetic.witness_like_scalars(n, 2222)):
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        torch.cuda.synchronize()
        got = rb.msm(device_ptr=d_sc.data_ptr(), npoints=n)
        kk = util.weighted_sum_mod_r(sc, start=1)
        assert util.affine_equal(oracle.g1_to_affine(got), oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), util.limbs(kk, 4))))
    rb.close()
    m = 1 << 21
    bases, k = _pseudo_random_bases(m, 0xBA5E6)
    sc = synthetic.random_fr_integers(m, 0x5CA1B)
    rb = RegisteredBases(bases, tables=13, window_bits=20)
    want = oracle.g1_to_affine(oracle.g1_mul(util.g1_generator_affine(), _inner_product_mod_r(oracle.fr_op("from_bigint", sc), k)[0]))
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc)), want)
    rb.close()


# ------------------------------------------------------------------------------------------ G2
def _g2_bases(golden, n):
    from oracle import cpu as o

    g2 = golden["constants"]["g2"]
    gen = np.zeros(1, dtype=o.G2_AFFINE)
    gen["x"] = g2["G2_GENERATOR_X_C0_MONT"] + g2["G2_GENERATOR_X_C1_MONT"]
    gen["y"] = g2["G2_GENERATOR_Y_C0_MONT"] + g2["G2_GENERATOR_Y_C1_MONT"]
    proj = np.zeros(n, dtype=o.G2_PROJECTIVE)
    for i in range(n):
        proj[i] = o.g2_mul(gen, util.limbs(3 * i + 1, 4))[0]
    return o.g2_to_affine(proj)


@pytest.mark.parametrize("n", [1, 7, 33, 100, 1000, 5000])
def test_g2_msm_vs_oracle_standard_msm(golden, n):
    """G2 goes through `standard::msm` in the reference (variable_base/mod.rs:45-47, standard.rs:79-105)."""
    from snarkvm_amd.msm import msm_g2

    bases = _g2_bases(golden, n)
    sc = synthetic.random_fr_integers(n, 800 + n)
    if n >= 33:
        sc[1] = 0
        sc[2] = [1, 0, 0, 0]  # scalar == 1 takes the dedicated branch of standard.rs:48-53
        bases[5] = bases[4]  # duplicate base
        bases[6]["infinity"] = 1
    got = oracle.g2_to_affine(msm_g2(bases, sc))
    want = oracle.g2_to_affine(oracle.g2_msm(bases, sc, oracle.MSM_STANDARD))
    assert got.tobytes() == want.tobytes()


def test_g2_msm_2_16_vs_oracle():
    """BASELINE.json configs[4]'s G2 leg at its own size: 2^16 pairs (512 distinct multiples of the generator tiled, the shape
    of the reference's MSM benches), host-buffer call and registered 16-table call, both == `standard::msm` of the oracle."""
    from snarkvm_amd.msm import RegisteredBasesG2, msm_g2

    n = 1 << 16
    bases = synthetic.g2_points(n)
    sc = synthetic.random_fr_integers(n, 1616)
    sc[3] = 0
    sc[4] = [1, 0, 0, 0]
    want = oracle.g2_to_affine(oracle.g2_msm(bases.view(oracle.G2_AFFINE), sc, oracle.MSM_STANDARD)).tobytes()
    assert oracle.g2_to_affine(msm_g2(bases, sc)).tobytes() == want
    rg = RegisteredBasesG2(bases, tables=16)
    try:
        assert oracle.g2_to_affine(rg.msm(sc)).tobytes() == want
        res = rg.msm_batch([sc[:40000], sc])
        assert oracle.g2_to_affine(res[1:2]).tobytes() == want
        assert oracle.g2_to_affine(res[0:1]).tobytes() == oracle.g2_to_affine(oracle.g2_msm(bases.view(oracle.G2_AFFINE)[:40000], sc[:40000], oracle.MSM_STANDARD)).tobytes()
    finally:
        rg.close()


# ------------------------------------------------------------------------------------------ KZG10 commit
def test_kzg10_commit_matches_reference_formula(golden):
    """KZG10::commit (polycommit/kzg10/mod.rs:98-156) = msm(powers[lz..], to_bigint(coeffs[lz..])) + msm(gamma powers,
    to_bigint(blinding)); checked against the oracle's CPU MSMs + projective addition, after to_affine."""
    from snarkvm_amd import kzg10

    n = 3000
    powers_g = _srs(golden, n)                                   # real SRS powers of beta (tiled)
    gamma_g = oracle.g1_gen_bases(util.g1_generator_affine(), 11, 8)  # stand-in for powers_of_beta_times_gamma_g
    pw = kzg10.Powers(powers_g, gamma_g)
    coeffs = oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, 4242))
    coeffs[:37] = 0                                              # leading zeros are skipped (mod.rs:455-467)
    coeffs[100] = 0
    blind = oracle.fr_op("from_bigint", synthetic.random_fr_integers(4, 99))
    for hiding in (None, 2):
        comm, rand = kzg10.KZG10.commit(pw, coeffs, hiding, (lambda k: blind[:k]) if hiding is not None else None)
        want = oracle.g1_msm(powers_g[37:], oracle.fr_op("to_bigint", coeffs[37:]), oracle.MSM_BATCHED)
        if hiding is not None:
            # KZGRandomness::rand -> DensePolynomial::rand(hiding + 1): degree hiding + 1, hiding + 2 coefficients
            # (kzg10/data_structures.rs:351-356, fft/polynomial/dense.rs:120-127)
            assert rand.blinding_polynomial.shape[0] == hiding + 2
            want = oracle.g1_add(want, oracle.g1_msm(gamma_g[: hiding + 2], oracle.fr_op("to_bigint", blind[: hiding + 2]), oracle.MSM_BATCHED))
        assert util.affine_equal(oracle.g1_to_affine(comm), oracle.g1_to_affine(want))
        # device-side `From<Projective> for Affine` agrees with the oracle's normalisation
        assert util.affine_equal(kzg10.to_affine(comm), oracle.g1_to_affine(want))
    # zero polynomial and too-large degree
    z, _ = kzg10.KZG10.commit(pw, np.zeros((0, 4), dtype=np.uint64))
    assert kzg10.to_affine(z)["infinity"][0] == 1
    with pytest.raises(kzg10.PCError):
        kzg10.KZG10.commit(pw, np.zeros((n + 1, 4), dtype=np.uint64) + 1)
    pw.close()


# ------------------------------------------------------------------------------------------ Varuna-shaped replay
def test_varuna_proof_shaped_workload(golden):
    """BASELINE.json configs[3]: the MSM / NTT / polymul call pattern of one Varuna proof with credits.aleo
    transfer_private shapes (|R| = |C| = 2^16, |K| = 2^17), random data, every result checked
    against the oracle.  Commits go through the fused KZG10 path over SRS powers registered once."""
    from snarkvm_amd import kzg10

    lgR, lgK = 16, 17
    g = util.g1_generator_affine()
    powers_g = oracle.g1_gen_bases(g, 1, 1 << (lgK + 1))
    gamma_g = oracle.g1_gen_bases(g, 1 << 20, 4)
    pw = kzg10.Powers(powers_g, gamma_g)

    def rnd(n, seed):
        return oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, seed))

    def check_commit(coeffs, hiding, seed):
        blind = rnd(4, seed)
        comm, _ = kzg10.KZG10.commit(pw, coeffs, hiding, (lambda k: blind[:k]) if hiding is not None else None)
        want = oracle.g1_msm(powers_g[: coeffs.shape[0]], oracle.fr_op("to_bigint", coeffs), oracle.MSM_BATCHED)
        if hiding is not None:
            want = oracle.g1_add(want, oracle.g1_msm(gamma_g[: hiding + 2], oracle.fr_op("to_bigint", blind[: hiding + 2]), oracle.MSM_BATCHED))
        assert util.affine_equal(oracle.g1_to_affine(comm), oracle.g1_to_affine(want))

    def check_ntt(x, direction, kind=NTTType.Standard):
        y = x.copy()
        plugin.NTT(x.shape[0], y, NTTInputOutputOrder.NN, direction, kind)
        assert np.array_equal(y, oracle.ntt(x, oracle.ORDER_NN, direction, kind))
        return y

    # round 1: x_poly FFT + w iFFT at |C|, commit w (hiding)
    w_evals = rnd(1 << lgR, 1)
    w = check_ntt(w_evals, NTTDirection.Inverse)
    check_ntt(rnd(1 << lgR, 2), NTTDirection.Forward)
    check_commit(w, 1, 3)
    # round 2: z_a, z_b, z_c iFFT at |R|; h_0 ~ z_a * z_b (2 FFT + 1 iFFT at 2|R|); commit h_0
    za, zb, zc = (check_ntt(rnd(1 << lgR, 10 + i), NTTDirection.Inverse) for i in range(3))
    h0 = plugin.polymul(1 << (lgR + 1), [za, zb], [])
    assert np.array_equal(h0, oracle.polymul(lgR + 1, [za, zb]))
    check_commit(h0, None, 0)
    # round 3: per matrix iFFT at |C| + polymul at 2|C|; commit g_1 (hiding), h_1
    for m in range(3):
        t = check_ntt(rnd(1 << lgR, 20 + m), NTTDirection.Inverse)
        pm = plugin.polymul(1 << (lgR + 1), [t, za], [])
        assert np.array_equal(pm, oracle.polymul(lgR + 1, [t, za]))
    check_commit(rnd((1 << lgR) - 1, 30), 1, 31)
    check_commit(h0, None, 0)
    # round 4: per matrix 3 iFFT at |K| + polymul at 2|K|; commit g_a, g_b, g_c
    for m in range(3):
        a_ = check_ntt(rnd(1 << lgK, 40 + m), NTTDirection.Inverse)
        b_ = check_ntt(rnd(1 << lgK, 50 + m), NTTDirection.Inverse)
        check_ntt(rnd(1 << lgK, 60 + m), NTTDirection.Inverse, NTTType.Coset)
        if m == 0:
            pm = plugin.polymul(1 << (lgK + 1), [a_, b_], [])
            assert np.array_equal(pm, oracle.polymul(lgK + 1, [a_, b_]))
        check_commit(a_[: (1 << lgK) - 1], None, 0)
    # round 5 + openings: commits of the largest combined polynomials, pipelined as one batch
    polys = [rnd((1 << lgK) - 2, 70), rnd(1 << lgK, 71), rnd(1 << lgR, 72), rnd(1 << lgK, 73)]
    rb = RegisteredBases(powers_g, tables=4)
    got = rb.msm_batch(polys, montgomery=True)
    for i, p_ in enumerate(polys):
        want = oracle.g1_msm(powers_g[: p_.shape[0]], oracle.fr_op("to_bigint", p_), oracle.MSM_BATCHED)
        assert util.affine_equal(oracle.g1_to_affine(got[i : i + 1]), oracle.g1_to_affine(want))
    rb.close()
    pw.close()


def test_g1_sum_of_partial_results(golden):
    """snarkvm_hip_g1_sum: the combine step of a point-range-split MSM equals the unsplit MSM."""
    from snarkvm_amd.msm import g1_sum

    n = 5000
    bases = _srs(golden, n)
    scalars = synthetic.random_fr_integers(n, 8080)
    rb = RegisteredBases(bases, tables=16)
    want = oracle.g1_to_affine(oracle.g1_msm(bases, scalars))
    for world in (1, 2, 3, 8, 70):
        parts = []
        for r in range(world):
            lo, hi = batch.split_range(n, world, r)
            parts.append(rb.msm(scalars[lo:hi], offset=lo))
        total = g1_sum(np.concatenate(parts))
        assert util.affine_equal(oracle.g1_to_affine(total), want), world
    inf = g1_sum(np.zeros(0, dtype=G1_PROJECTIVE))
    assert oracle.g1_to_affine(inf)["infinity"][0] == 1
    # P + (-P) and P + P through the complete addition
    p = rb.msm(scalars[:10])
    neg = oracle.g1_msm(bases[:10], oracle.fr_op("to_bigint", oracle.fr_op("neg", oracle.fr_op("from_bigint", scalars[:10]))))
    assert oracle.g1_to_affine(g1_sum(np.concatenate([p, neg])))["infinity"][0] == 1
    assert util.affine_equal(oracle.g1_to_affine(g1_sum(np.concatenate([p, p]))), oracle.g1_to_affine(oracle.g1_add(p, p)))
    rb.close()


@pytest.mark.parametrize("tables,window_bits", [(16, 0), (4, 0), (15, 17)])
def test_g2_registered_tables_vs_oracle(golden, tables, window_bits):
    """Registered G2 bases with precomputed tables (no Horner chain) equal `standard::msm` on the same inputs."""
    from snarkvm_amd.msm import RegisteredBasesG2

    n = 3000
    bases = _g2_bases(golden, n)
    bases[7]["infinity"] = 1
    sc = synthetic.random_fr_integers(n, 5150)
    sc[0] = 0
    sc[1] = util.limbs(1, 4)
    sc[2] = util.limbs(pyref.R_MOD - 1, 4)
    sc[9] = sc[10]
    rb = RegisteredBasesG2(bases, tables=tables, window_bits=window_bits)
    from oracle import cpu as o

    want = o.g2_to_affine(o.g2_msm(bases, sc))
    assert o.g2_to_affine(rb.msm(sc)).tobytes() == want.tobytes()
    if window_bits:
        assert o.g2_to_affine(rb.msm(sc, window_bits=window_bits)).tobytes() == want.tobytes()  # wide path over Fq2
    want2 = o.g2_to_affine(o.g2_msm(bases[100:433], sc[:333]))
    assert o.g2_to_affine(rb.msm(sc[:333], offset=100)).tobytes() == want2.tobytes()
    rb.close()


@pytest.mark.parametrize("tables,window_bits", [(17, 15), (16, 16), (19, 14), (22, 12), (1, 0)])
def test_msm_of_repeated_points_repeats_and_matches_the_oracle(tables, window_bits):
    """1 024 bases that are SIXTEEN distinct points tiled (the reference's MSM benches tile a small set too, benches/msm/variable_base.rs:29-32): equal
    partial sums meet all over the fold / bit-plane trees, so every cooperative addition takes its equal-x fallback (P + P, P - P: the out-of-line plain
    law) again and again - the path a random vector exercises once in 2^377.  Twelve calls each: the buckets fill in another order every time (scatter
    atomics), every call must give `standard::msm`'s sum.  Round 6 found this path returning another wrong sum on every call for G2 when the compiler's
    out-of-line fallbacks came back with registers of the caller overwritten (snarkvm_amd/build.py; the Fq2 tail kernels now hold no call: msm.hip.h
    TAIL_FLAGGED) - invisible to the single-shot parity tests over distinct points."""
    from snarkvm_amd.msm import RegisteredBases, RegisteredBasesG2

    n = 1024
    sc = synthetic.random_fr_integers(n, 1600 + tables)
    g2 = synthetic.g2_points(n, distinct=16)
    want2 = oracle.g2_to_affine(oracle.g2_msm(g2.view(oracle.G2_AFFINE), sc, oracle.MSM_STANDARD)).tobytes()
    rg = RegisteredBasesG2(g2, tables=tables, window_bits=window_bits)
    try:
        for rep in range(12):
            assert oracle.g2_to_affine(rg.msm(sc)).tobytes() == want2, ("g2", rep)
    finally:
        rg.close()
    g1 = np.tile(oracle.g1_gen_bases(util.g1_generator_affine(), 5, 16), n // 16)
    want1 = oracle.g1_to_affine(oracle.g1_msm(g1, sc)).tobytes()
    rb = RegisteredBases(g1, tables=tables, window_bits=window_bits)
    try:
        for rep in range(12):
            assert oracle.g1_to_affine(rb.msm(sc)).tobytes() == want1, ("g1", rep)
    finally:
        rb.close()


def test_repeated_points_at_sizes_that_reach_the_other_exceptional_paths():
    """The same shape at sizes whose buckets hold many entries, so that the accumulate kernels (G1: the lazy law's exact exceptional addition; G2: the lane-pair
    doubling), the reduce rounds and the bucket merge meet equal points too: 2^18 G1 pairs over 16 distinct points (host buffers, then registered tables), 2^14
    G2 pairs over 4 distinct points (host buffers, 17 x 15 tables, a fused batch of three) - three calls each, every one against the oracle."""
    from snarkvm_amd.msm import RegisteredBases, RegisteredBasesG2, msm_g2

    n = 1 << 18
    g1 = np.tile(oracle.g1_gen_bases(util.g1_generator_affine(), 9, 16), n // 16)
    sc = synthetic.random_fr_integers(n, 181818)
    want1 = oracle.g1_to_affine(oracle.g1_msm(g1, sc)).tobytes()
    for rep in range(3):
        assert oracle.g1_to_affine(plugin.msm(g1, sc)).tobytes() == want1, ("g1 host", rep)
    for tables, wb in ((1, 0), (17, 15), (13, 20)):
        rb = RegisteredBases(g1, tables=tables, window_bits=wb)
        try:
            for rep in range(3):
                assert oracle.g1_to_affine(rb.msm(sc)).tobytes() == want1, ("g1", tables, rep)
            if tables == 17:  # a fused group of three (768 fold workgroups: the 128-thread fold of tuning fold_mid)
                res = rb.msm_batch([sc, sc[: n // 2], sc])
                assert oracle.g1_to_affine(res[0:1]).tobytes() == want1 and oracle.g1_to_affine(res[2:3]).tobytes() == want1
                assert oracle.g1_to_affine(res[1:2]).tobytes() == oracle.g1_to_affine(oracle.g1_msm(g1[: n // 2], sc[: n // 2])).tobytes()
        finally:
            rb.close()
    m = 1 << 14
    g2 = synthetic.g2_points(m, distinct=4)
    sc2 = synthetic.random_fr_integers(m, 141414)
    want2 = oracle.g2_to_affine(oracle.g2_msm(g2.view(oracle.G2_AFFINE), sc2, oracle.MSM_STANDARD)).tobytes()
    for rep in range(3):
        assert oracle.g2_to_affine(msm_g2(g2, sc2)).tobytes() == want2, ("g2 host", rep)
    rg = RegisteredBasesG2(g2, tables=17, window_bits=15)
    try:
        for rep in range(3):
            assert oracle.g2_to_affine(rg.msm(sc2)).tobytes() == want2, ("g2", rep)
        res = rg.msm_batch([sc2, sc2[: m // 2], sc2])
        assert oracle.g2_to_affine(res[0:1]).tobytes() == want2 and oracle.g2_to_affine(res[2:3]).tobytes() == want2
        want_half = oracle.g2_to_affine(oracle.g2_msm(g2.view(oracle.G2_AFFINE)[: m // 2], sc2[: m // 2], oracle.MSM_STANDARD)).tobytes()
        assert oracle.g2_to_affine(res[1:2]).tobytes() == want_half
    finally:
        rg.close()


@pytest.mark.parametrize("m,hb", [(7, 7), (7, 6), (8, 7)])
def test_g2_tail_kernels_give_the_same_sums_on_every_launch(m, hb):
    """snarkvm_hip_devtest_g2_tail_repeat: the Fq2 fold and bit-plane kernels over ONE fixed set of per-bucket lists built from a few repeated points (equal-x
    fallbacks everywhere), launched 40 times at every workgroup size and with the cooperative-addition switches on and off: the same group elements every time."""
    import ctypes

    from snarkvm_amd import _lib

    L = _lib.lib()
    pts = synthetic.g2_points(512)
    for threads in (64, 128, 256):
        for hex2 in (0, 1):
            for quads in (0, 3):
                if threads == 64 and quads:
                    continue
                rep = np.zeros(10, dtype=np.uint32)
                _lib.check(L.snarkvm_hip_devtest_g2_tail_repeat(ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(512), m, hb, threads, 256 if quads else 128, hex2, quads, 40,
                                                                ctypes.c_void_p(rep.ctypes.data)))
                assert not rep[:4].any(), (threads, hex2, quads, rep.tolist())
                assert rep[8] > 0, "the repeated points of this input must send some outputs through the fix kernel"


def test_ffi_base_cache_is_transparent(tmp_path):
    """SNARKVM_HIP_BASE_CACHE (opt-in extension; `snarkvm_msm` is stateless without it): the FFI reusing device copies of base
    ranges it has seen - same results for repeated calls, sub-slices with an offset, a superseding bigger range, and memory
    that changed in place.  Registration only ever reads the slice the registering call passed."""
    import subprocess
    import sys

    script = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from oracle import cpu as oracle
from snarkvm_amd import plugin, synthetic
from tests import util
n = 20000
bases = oracle.g1_gen_bases(util.g1_generator_affine(), 3, n)
sc = synthetic.random_fr_integers(n, 2468)
def check(b, s):
    got = oracle.g1_to_affine(plugin.msm(b, s))
    assert util.affine_equal(got, oracle.g1_to_affine(oracle.g1_msm(b, s))), "mismatch"
check(bases[:9000], sc[:9000])          # first sighting: remembered, uncached path
check(bases[:9000], sc[:9000])          # second sighting: registers [0, 9000), served from HBM
check(bases[:9000], sc[:9000])          # hit
check(bases[100:5100], sc[:5000])       # hit with an offset that is not a sampled position
check(bases[1:1100], sc[:1099])         # short slice between sampled positions: still >= 16 samples inside
check(bases, sc)                        # bigger range supersedes the first one (first sighting again)
check(bases[4096:12000], sc[:7904])     # sub-slice of a range that is not registered yet: stateless path (only this slice may be read)
check(bases, sc)                        # second sighting of exactly the big range: registered from this call's memory
check(bases[4096:12000], sc[:7904])     # hit starting on a sampled point
check(bases, sc)                        # hit on the whole range
bases[8192] = bases[1]                  # the memory changes in place at a sampled position
check(bases, sc)                        # detected -> dropped, uncached path
check(bases, sc)                        # registered again with the new content
check(bases, sc)
bases[8191] = bases[2]                  # an unsampled position: documented limitation, so continue through a fresh array
fresh = bases.copy()
check(fresh, sc)
check(fresh, sc)
check(fresh[37:12345], sc[:12308])
print("CACHE_OK")
''' % util.ROOT
    env = dict(os.environ, SNARKVM_HIP_BASE_CACHE="4")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=600)
    assert "CACHE_OK" in r.stdout, r.stdout + r.stderr


def test_ffi_msm_is_stateless_by_default():
    """the callee must not retain the caller's pointers.  Without SNARKVM_HIP_BASE_CACHE in the environment
    `snarkvm_msm` re-reads its operands on every call: a base changed in place between two calls (at a position the opt-in
    cache would not sample) changes the result accordingly."""
    if os.environ.get("SNARKVM_HIP_BASE_CACHE", "0") not in ("", "0"):
        pytest.skip("the opt-in base cache is enabled in this environment")
    n = 20000
    bases = oracle.g1_gen_bases(util.g1_generator_affine(), 3, n)
    sc = synthetic.random_fr_integers(n, 1357)
    for _ in range(3):
        _check(bases, sc, VariableBase.msm(bases, sc))
    bases[8191] = bases[2]
    bases[8192] = bases[1]
    for _ in range(2):
        _check(bases, sc, VariableBase.msm(bases, sc))


def test_msm_randomized_configurations(golden):
    """Differential sweep over deterministic pseudo-random configurations: size, base-table geometry, forced window width
    and scalar distribution (uniform / tiny / sparse / all-equal / near r), registered and plain FFI, against `batched::msm`."""
    rng = np.random.default_rng(20240924)
    pool = _srs(golden, 30000)
    geometries = [(1, 0), (4, 0), (16, 0), (15, 17), (13, 20), (12, 22)]
    for case in range(24):
        n = int(rng.choice([1, 2, 17, 255, 1000, 4097, 12345, 30000]))
        off = int(rng.integers(0, 30000 - n + 1))
        tables, bits = geometries[int(rng.integers(0, len(geometries)))]
        kind = int(rng.integers(0, 5))
        sc = synthetic.random_fr_integers(n, 5000 + case)
        if kind == 1:
            sc[:, 1:] = 0
            sc[:, 0] &= np.uint64(0xFFFF)
        elif kind == 2:
            sc[rng.random(n) < 0.7] = 0
        elif kind == 3:
            sc[:] = sc[0]
        elif kind == 4:
            sc[:] = util.limbs(pyref.R_MOD - 1 - case, 4)
        bases = pool[off : off + n]
        want = oracle.g1_to_affine(oracle.g1_msm(bases, sc))
        rb = RegisteredBases(pool, tables=tables, window_bits=bits)
        forced = bits if (bits and case % 2 == 0) else 0
        got = rb.msm(sc, offset=off, window_bits=forced)
        assert util.affine_equal(oracle.g1_to_affine(got), want), (case, n, off, tables, bits, kind, forced)
        rb.close()
        if case % 4 == 0:
            assert util.affine_equal(oracle.g1_to_affine(VariableBase.msm(bases, sc)), want), ("ffi", case, n, kind)


@pytest.mark.parametrize("n", [5000, 70000, 200000])
def test_msm_skewed_scalars_every_tail_shape(n):
    """The tail kernels take whatever partial sums the accumulate phase leaves (no reduce round below 2^22 digit entries).
    Heavily skewed scalar vectors - all equal, two values, half ones, 90 % tiny - through the three shapes of the tail:
    the unfolded bit planes of a table-less MSM (n <= 2^17: c <= 11), the one-wave fold of a table-less c = 16 MSM (200 000),
    and the 256-thread fold over 16 registered tables; bases (i + 1) G, expected = (sum s_i (i + 1)) G by the oracle."""
    G = util.g1_generator_affine()
    bases = oracle.g1_gen_bases(G, 1, n)
    rng = np.random.default_rng(n)
    uni = synthetic.random_fr_integers(n, 4000 + n)
    vecs = {"all equal": np.tile(uni[:1], (n, 1)), "two values": uni[rng.integers(0, 2, n)]}
    half = uni.copy()
    half[rng.random(n) < 0.5] = [1, 0, 0, 0]
    vecs["half ones"] = half
    tiny = np.zeros_like(uni)
    tiny[:, 0] = rng.integers(0, 256, n, dtype=np.uint64)
    keep = rng.random(n) < 0.1
    tiny[keep] = uni[keep]
    vecs["90 % tiny"] = tiny
    rb = RegisteredBases(bases, tables=16)
    try:
        for name, sc in vecs.items():
            sc = np.ascontiguousarray(sc)
            want = oracle.g1_to_affine(oracle.g1_mul(G, util.limbs(util.weighted_sum_mod_r(sc, start=1), 4)))
            fresh = bases.copy()  # a host range the base cache has not seen: the table-less path
            assert util.affine_equal(oracle.g1_to_affine(plugin.msm(fresh, sc)), want), (name, "table-less")
            assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc)), want), (name, "16 tables")
    finally:
        rb.close()


def test_extension_abi_rejects_bad_arguments(golden):
    """Every extension entry point answers a malformed request with a non-zero RustError (never a crash, never a silent
    result): the caller's fallback logic depends on it (variable_base/mod.rs:39-43)."""
    L = _lib.lib()
    bases = _srs(golden, 64)
    h = ctypes.c_void_p()

    def err(e):
        with pytest.raises(_lib.HipError):
            _lib.check(e)

    P = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    err(L.snarkvm_hip_register_bases_windowed(ctypes.byref(h), P(bases), ctypes.c_size_t(64), ctypes.c_size_t(104), 0, 11, 22))   # 11 * 22 < 254
    err(L.snarkvm_hip_register_bases_windowed(ctypes.byref(h), P(bases), ctypes.c_size_t(64), ctypes.c_size_t(104), 0, 12, 24))   # window > 23
    err(L.snarkvm_hip_register_bases_windowed(ctypes.byref(h), P(bases), ctypes.c_size_t(64), ctypes.c_size_t(104), 0, 16, 23))   # 368 digit bits > 288
    err(L.snarkvm_hip_register_bases_windowed(ctypes.byref(h), P(bases), ctypes.c_size_t(64), ctypes.c_size_t(104), 0, 14, 23))   # 322 digit bits > 288
    err(L.snarkvm_hip_register_bases_tables(ctypes.byref(h), P(bases), ctypes.c_size_t(64), ctypes.c_size_t(104), 0, 3))         # not a power of two
    err(L.snarkvm_hip_register_bases_tables(ctypes.byref(h), P(bases), ctypes.c_size_t(64), ctypes.c_size_t(100), 0, 4))         # stride < 104
    rb = RegisteredBases(bases, tables=4)
    sc = synthetic.random_fr_integers(64, 1)
    out = np.zeros(1, dtype=G1_PROJECTIVE)
    err(L.snarkvm_hip_msm_registered(P(out), rb._h, ctypes.c_size_t(1), ctypes.c_size_t(64), P(sc), 0, 0))     # range past the end
    err(L.snarkvm_hip_msm_registered(P(out), rb._h, ctypes.c_size_t(0), ctypes.c_size_t(64), P(sc), 0, 24))    # window_bits > 23
    rb.close()
    x = np.zeros((8, 4), dtype=np.uint64)
    err(L.snarkvm_hip_fr_vec_op(99, P(x), P(x), P(x), None, None, ctypes.c_size_t(8), 0))                      # unknown op
    err(L.snarkvm_hip_fr_vec_op(4, P(x), P(x), None, None, None, ctypes.c_size_t(8), 0))                       # scale without a scalar
    err(L.snarkvm_hip_fr_divide_by_vanishing(P(x), P(x), P(x), ctypes.c_size_t(8), ctypes.c_size_t(0), 0))    # empty domain
    err(L.snarkvm_hip_fr_lagrange_coefficients(P(x), ctypes.c_uint32(31), P(x), 0))                            # lg > 30
    err(L.snarkvm_hip_g1_group_ntt(P(out), ctypes.c_uint32(25), 0))                                            # lg > 24
    err(L.snarkvm_hip_ntt_device(ctypes.c_void_p(0x1000), ctypes.c_uint32(27), 0, 0, 0))                       # lg > 26: caller falls back
    err(L.snarkvm_hip_ntt_device(ctypes.c_void_p(0x1000), ctypes.c_uint32(4), 7, 0, 0))                        # bad enum
    one = (ctypes.c_void_p * 1)(0x1000)
    err(L.snarkvm_hip_ntt_device_batch(one, ctypes.c_size_t(1), ctypes.c_uint32(27), 0, None, None))          # lg > 26
    err(L.snarkvm_hip_ntt_device_batch(None, ctypes.c_size_t(1), ctypes.c_uint32(4), 0, None, None))           # null list
    err(L.snarkvm_hip_ntt_device_batch(one, ctypes.c_size_t(1), ctypes.c_uint32(4), 0, None, None))            # not a device pointer of a device in use
    err(L.snarkvm_hip_g1_serialize(P(x), P(bases), ctypes.c_size_t(1), ctypes.c_size_t(96), 0))                # stride < 104


def test_bench_two_rank_path_on_one_gpu():
    """bench.py's N > 1 code path (rendezvous, barriers, max-over-ranks timing, rank-0 JSON) with two ranks sharing this
    GPU over gloo (RCCL refuses two ranks on one device; the collective semantics are the same)."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, SNARKVM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--lg-msm", "16", "--lg-ntt", "16", "--ntt-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=util.ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 2 * (1 << 16) / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the self-check of a scaling point: the sum of the ranks' own rates beside `value`, and the two ranks sharing this GPU are flagged
    assert d["predicted_value"] >= d["value"] * 0.999 and 0 < d["value_over_predicted"] <= 1.001
    if all(r.get("uuid") for r in d["rank_devices"]):
        assert len(d["duplicate_devices"]) == 1 and sorted(d["duplicate_devices"][0]["ranks"]) == [0, 1]


def test_bench_proofs64_two_rank_lockstep_on_one_gpu():
    """bench.py --workload proofs64 with two ranks sharing this GPU over gloo: the proofs are sharded over the ranks (strong scaling),
    every rank replays its share in lock step and by concurrent callers, the modes are compared, rank 0 checks one whole proof
    against the oracle and prints the one JSON line with every rank's own time."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, SNARKVM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29534",
           os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--workload", "proofs64", "--proofs", "8", "--proof-group", "4", "--proof-workers", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=util.ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "strong" and len(d["rank_ms_per_proof"]) == 2
    assert "one_proof_vs_oracle" in d["checks"] and "lockstep_vs_callers" in d["checks"]
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 8e-3)) / d["value"] < 1e-6


def test_bench_proof1_two_rank_on_one_gpu():
    """bench.py --workload proof1 with two ranks sharing this GPU over gloo (BASELINE.json configs[3] per rank, weak scaling): every rank
    proves its own 32 proofs one at a time, compares each with the serial replay; rank 0 checks two whole proofs against the oracle and
    prints the one JSON line with every rank's own time."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, SNARKVM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29535",
           os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--workload", "proof1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=util.ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 32 and d["scaling"] == "weak" and len(d["rank_ms_per_proof"]) == 2
    assert "proofs_vs_oracle" in d["checks"] and "every_proof_vs_serial_replay" in d["checks"]
    assert d["latency"]["workspace_growth_in_timed_region"]["device_allocations"] == 0
    assert d["value"] > 0 and abs(d["value"] - 2 * 32 / (d["ms_per_step"] * 32e-3)) / d["value"] < 1e-6
