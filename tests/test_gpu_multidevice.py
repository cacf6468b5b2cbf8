"""Multi-device paths of the C ABI on a one-GPU box: the same GPU listed twice (SNARKVM_HIP_DEVICES=0,0) gives two independent
logical devices - own streams, workspaces and base replicas - so the point-range split of one MSM, the replication of
registered bases, the batch fan-out and the lane tokens of concurrent callers all run their real code.  Each scenario is a
subprocess because the device set is fixed at the first compute call of a process."""
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import ctypes, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, %r)
import torch
from oracle import cpu as oracle
from oracle import pyref
from snarkvm_amd import _lib, msm, plugin, synthetic
from snarkvm_amd.layout import NTTDirection, NTTInputOutputOrder, NTTType
from tests import util

assert msm.num_devices() == 2, msm.num_devices()
G = util.g1_generator_affine()
n = (1 << 19) + 321
bases = oracle.g1_gen_bases(G, 1, n)
sc = synthetic.random_fr_integers(n, 777)
def closed(s, start=1):
    return oracle.g1_to_affine(oracle.g1_mul(G, util.limbs(util.weighted_sum_mod_r(s, start=start), 4)))
def eq(got, want):
    return util.affine_equal(oracle.g1_to_affine(got), want)
want = closed(sc)
# 1. plain FFI: cut into two point-range chunks, one per logical device, combined on the host
assert eq(plugin.msm(bases, sc), want), "ffi split"
# 2. second and third sighting of the same host range: registered on both devices (base cache), split by point range
assert eq(plugin.msm(bases, sc), want), "ffi cached 1"
assert eq(plugin.msm(bases, sc), want), "ffi cached 2"
assert eq(plugin.msm(bases[5:300005], sc[:300000]), closed(sc[:300000], start=6)), "ffi cached slice"
# 3. registered bases are replicated; host scalars split over both replicas, device scalars run where they live
rb = msm.RegisteredBases(bases, tables=16)
assert eq(rb.msm(sc), want), "registered host scalars"
d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
torch.cuda.synchronize()
for _ in range(3):  # the owner lookup rotates over the logical devices that share the physical GPU
    assert eq(rb.msm(device_ptr=d_sc.data_ptr(), npoints=n), want), "registered device scalars"
# 4. batch fan-out: instance k on logical device k mod 2, several lanes each; ragged sizes and offsets
sizes = [1000, 70000, 1, 33333, 0, 262144, 4097]
offs = [0, 11, 500000, 7, 3, 1000, 99]
res = rb.msm_batch([sc[:m] for m in sizes], offsets=offs)
for k, (m, o) in enumerate(zip(sizes, offs)):
    if m == 0:
        assert oracle.g1_to_affine(res[k:k+1])["infinity"][0] == 1
    else:
        assert eq(res[k:k+1], closed(sc[:m], start=o + 1)), ("batch", k)
res = rb.msm_batch(device_ptrs=[d_sc.data_ptr()] * 5, npoints=[n, 5000, n, 123, 65536])
for k, m in enumerate([n, 5000, n, 123, 65536]):
    assert eq(res[k:k+1], closed(sc[:m])), ("device batch", k)
# KZG10-shaped call with two base ranges, host scalars (split path with a range boundary inside a part)
out = np.zeros(1, dtype=oracle.G1_PROJECTIVE)
n0, n1, off1 = 400000, 124000, 100
_lib.check(_lib.lib().snarkvm_hip_msm_registered_ex(ctypes.c_void_p(out.ctypes.data), rb._h, ctypes.c_size_t(0), ctypes.c_size_t(n0), ctypes.c_size_t(off1),
                                                   ctypes.c_size_t(n1), ctypes.c_void_p(sc.ctypes.data), 0, 0, 0))
k = (util.weighted_sum_mod_r(sc[:n0], start=1) + util.weighted_sum_mod_r(sc[n0:n0 + n1], start=off1 + 1)) %% pyref.R_MOD
assert eq(out, oracle.g1_to_affine(oracle.g1_mul(G, util.limbs(k, 4)))), "two-range split"
rb.close()
# 5. concurrent callers (rayon workers): 8 threads, NTTs and small MSMs at once on whatever lane / device is free
xs = [oracle.fr_op("from_bigint", synthetic.random_fr_integers(1 << 13, 900 + t)) for t in range(8)]
def work(t):
    y = xs[t].copy()
    plugin.NTT(1 << 13, y, NTTInputOutputOrder.NN, NTTDirection.Forward, NTTType.Standard)
    ok = np.array_equal(y, oracle.ntt(xs[t]))
    m = 3000 + 100 * t
    ok = ok and eq(plugin.msm(bases[t:t + m], sc[:m]), closed(sc[:m], start=t + 1))
    return ok
with ThreadPoolExecutor(8) as ex:
    assert all(ex.map(work, range(8))), "concurrent callers"
# 6. G2: registered on both devices, batch over devices
from snarkvm_amd.layout import G2_AFFINE
raw = open(%r, "rb").read()
g2 = np.zeros(1, dtype=G2_AFFINE)
_lib.check(_lib.lib().snarkvm_hip_g2_deserialize(ctypes.c_void_p(g2.ctypes.data), ctypes.c_char_p(raw), ctypes.c_size_t(1), 0))
g2b = np.concatenate([g2] * 600)
g2sc = synthetic.random_fr_integers(600, 31337)
rg = msm.RegisteredBasesG2(g2b, tables=16)
want2 = oracle.g2_to_affine(oracle.g2_msm(g2b, g2sc)).tobytes()
res = rg.msm_batch([g2sc, g2sc[:100], g2sc])
assert oracle.g2_to_affine(res[0:1]).tobytes() == want2 and oracle.g2_to_affine(res[2:3]).tobytes() == want2
assert oracle.g2_to_affine(res[1:2]).tobytes() == oracle.g2_to_affine(oracle.g2_msm(g2b[:100], g2sc[:100])).tobytes()
rg.close()
print("MULTI_OK")
'''


def test_two_logical_devices_end_to_end():
    env = dict(os.environ, SNARKVM_HIP_DEVICES="0,0", SNARKVM_HIP_BASE_CACHE="16")  # the opt-in base cache: its multi-device path stays covered
    script = SCRIPT % (util.ROOT, os.path.join(util.ROOT, "tests", "golden", "beta_h_g2.bin"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=1200, cwd=util.ROOT)
    assert "MULTI_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_device_set_is_fixed_after_first_use():
    script = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from snarkvm_amd import _lib, msm
msm.set_devices([0])
assert msm.num_devices() == 1
msm.set_devices([0])            # the identical list is accepted
try:
    msm.set_devices([0, 0])     # a different one is refused once the runtime exists
    raise SystemExit("accepted a second device set")
except _lib.HipError:
    pass
try:
    import ctypes
    _lib.check(_lib.lib().snarkvm_hip_set_devices(None, ctypes.c_size_t(0)))
    raise SystemExit("accepted an empty device list")
except _lib.HipError:
    pass
print("SET_OK")
''' % util.ROOT
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, cwd=util.ROOT)
    assert "SET_OK" in r.stdout, r.stdout + r.stderr


RING_SCRIPT = r'''
import ctypes, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import cpu as oracle
from oracle import pyref
from snarkvm_amd import _lib, msm, plugin, synthetic
from snarkvm_amd.layout import G2_AFFINE
from tests import util

assert msm.num_devices() == 1
G = util.g1_generator_affine()
n = (1 << 19) + 4099
bases = oracle.g1_gen_bases(G, 1, n)
sc = synthetic.random_fr_integers(n, 4242)
sc[5] = 0
sc[6] = [1, 0, 0, 0]
def closed(s, start=1):
    return oracle.g1_to_affine(oracle.g1_mul(G, util.limbs(util.weighted_sum_mod_r(s, start=start), 4)))
def eq(got, want):
    return util.affine_equal(oracle.g1_to_affine(got), want)
want = closed(sc)
# first sighting: 9 chunks of <= 2^16 pairs through the three-lane ring (uploader thread + compute thread)
assert eq(plugin.msm(bases, sc), want), "chunk ring, uncached"
print("uncached ok")
# second sighting registers the range; from then on only the scalars cross PCIe, in 3 chunks of <= 2^18
assert eq(plugin.msm(bases, sc), want), "scalar chunks 1"
assert eq(plugin.msm(bases, sc), want), "scalar chunks 2"
print("cached ok")
assert eq(plugin.msm(bases[17:17 + (1 << 19)], sc[: 1 << 19]), closed(sc[: 1 << 19], start=18)), "scalar chunks, slice"
print("slice ok")
# explicit handle, two base ranges with the boundary inside a chunk
rb = msm.RegisteredBases(bases, tables=16)
out = np.zeros(1, dtype=oracle.G1_PROJECTIVE)
n0, n1, off1 = 300001, 228000, 77  # n0 + n1 <= len(sc)
_lib.check(_lib.lib().snarkvm_hip_msm_registered_ex(ctypes.c_void_p(out.ctypes.data), rb._h, ctypes.c_size_t(3), ctypes.c_size_t(n0), ctypes.c_size_t(off1),
                                                   ctypes.c_size_t(n1), ctypes.c_void_p(sc.ctypes.data), 0, 0, 0))
k = (util.weighted_sum_mod_r(sc[:n0], start=4) + util.weighted_sum_mod_r(sc[n0:n0 + n1], start=off1 + 1)) %% pyref.R_MOD
assert eq(out, oracle.g1_to_affine(oracle.g1_mul(G, util.limbs(k, 4)))), "two ranges over scalar chunks"
print("two ranges ok")
rb.close()
# G2 through the same ring (the reference's symbol shape, host buffers)
raw = open(%r, "rb").read()
g2 = np.zeros(1, dtype=G2_AFFINE)
_lib.check(_lib.lib().snarkvm_hip_g2_deserialize(ctypes.c_void_p(g2.ctypes.data), ctypes.c_char_p(raw), ctypes.c_size_t(1), 0))
m2 = (1 << 19) + 5
g2b = np.concatenate([g2] * m2)
g2sc = synthetic.random_fr_integers(m2, 99)
ksum = 0
for limb in range(4):
    ksum += int(np.sum(g2sc[:, limb].astype(object))) << (64 * limb)
got = oracle.g2_to_affine(msm.msm_g2(g2b, g2sc))
want2 = oracle.g2_to_affine(oracle.g2_msm(g2[:1], np.array([[(ksum %% pyref.R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64)))
assert got.tobytes() == want2.tobytes(), "G2 chunk ring"
print("RING_OK")
'''


def test_chunk_ring_single_device():
    """The pipelined host-buffer paths at test size: tuning msm_chunk_lg=16 cuts a 2^19 `snarkvm_msm` into chunks over the
    three-lane ring (runtime.hip.h::lane_ring_run), scalar_chunk_lg=18 cuts the cached call's scalars into 3."""
    env = dict(os.environ, SNARKVM_HIP_DEVICES="0", SNARKVM_HIP_TUNING="msm_chunk_lg=16,scalar_chunk_lg=18", SNARKVM_HIP_BASE_CACHE="16")
    script = RING_SCRIPT % (util.ROOT, os.path.join(util.ROOT, "tests", "golden", "beta_h_g2.bin"))
    r = subprocess.run([sys.executable, "-u", "-c", script], capture_output=True, text=True, env=env, timeout=1200, cwd=util.ROOT)
    assert "RING_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


RAMP_SCRIPT = r'''
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, %r)
from oracle import cpu as oracle
from snarkvm_amd import _lib, msm, plugin
from snarkvm_amd.layout import G1_AFFINE
from tests import util

G = util.g1_generator_affine()
nmax = (1 << 23) + 77
buf = torch.empty(nmax * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
_lib.check(_lib.lib().snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax)))
bases = buf.cpu().numpy().view(G1_AFFINE)
sc = np.random.default_rng(31337).integers(0, 1 << 64, size=(nmax, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 60) - 1)  # < 2^252 < r: canonical integers
sc[3] = 0
def closed(n):
    return oracle.g1_to_affine(oracle.g1_mul(G, util.limbs(util.weighted_sum_mod_r(sc[:n], start=1), 4)))
# snarkvm_msm, host bases + host scalars: 2 chunks (front ramp only), 2 chunks of 1.5 * 2^20, 5 chunks (ramp + taper)
for n in ((1 << 20) + 12345, 3 * (1 << 20) + 5, 5 * (1 << 20) + 77):
    assert util.affine_equal(oracle.g1_to_affine(plugin.msm(bases[:n], sc[:n])), closed(n)), n
    print("ok", n)
# registered bases (13 tables x 20-bit windows) + host scalars: one upload, two geometric scalar chunks, three - all into one bucket sink
rb = msm.RegisteredBases(device_ptr=buf.data_ptr(), npoints=nmax, tables=13, window_bits=20)
for n in ((1 << 21) + 3, 5 * (1 << 20) + 77, nmax):
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc[:n])), closed(n)), ("registered", n)
    print("registered ok", n)
rb.close()
print("RAMP_OK")
'''


@pytest.mark.parametrize("tuning", ["ramp=3", "ramp=1,scalar_geo=2", "ramp=0,scalar_geo=0", "taper=0", "ramp=3,ring_lanes=2,scalar_chunk_lg=20,scalar_geo=0"])
def test_chunk_ramp_and_taper_uncached(tuning):
    """`snarkvm_msm` over host buffers with the default 2^21-pair chunks (no base cache: every call uploads): the first chunk is cut into a
    ramp (tuning ramp), the last into 1/2, 1/4, 1/4 (taper), every chunk adds into the shared bucket sink - same group element whatever
    the cut (runtime.hip.h::msm_host_chunked).  Then host scalars over registered bases: equal or geometric scalar chunks into one
    sink with chained merges (api.hip::msm_registered_host_scalars)."""
    env = dict(os.environ, SNARKVM_HIP_DEVICES="0", SNARKVM_HIP_TUNING=tuning, SNARKVM_HIP_BASE_CACHE="0")
    r = subprocess.run([sys.executable, "-u", "-c", RAMP_SCRIPT % util.ROOT], capture_output=True, text=True, env=env, timeout=600, cwd=util.ROOT)
    assert "RAMP_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


# ---------------------------------------------------------------------------------------------------------------------------
# The A/B switches DESIGN.md quotes measurements for select other kernels / launch shapes for the same mathematics: every one of
# them must return bit-identical results.  One subprocess per setting (the switches are read once per process).
AB_SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from oracle import cpu as oracle
from snarkvm_amd import msm, plugin, synthetic
from snarkvm_amd.layout import NTTDirection, NTTInputOutputOrder, NTTType
from tests import util

G = util.g1_generator_affine()
def closed(s, start=1):
    return oracle.g1_to_affine(oracle.g1_mul(G, util.limbs(util.weighted_sum_mod_r(s, start=start), 4)))
# NTT at proof sizes, all four transforms (the tile rule, tuning ntt_min_tiles)
for lg in (12, 14, 15, 16, 17):
    x = oracle.fr_op("from_bigint", synthetic.random_fr_integers(1 << lg, 4100 + lg))
    for d in (NTTDirection.Forward, NTTDirection.Inverse):
        for t in (NTTType.Standard, NTTType.Coset):
            y = x.copy()
            plugin.NTT(1 << lg, y, NTTInputOutputOrder.NN, d, t)
            assert np.array_equal(y, oracle.ntt(x, oracle.ORDER_NN, d, t)), ("ntt", lg, d, t)
# p / (X - z) and p(z) (tuning horner2: the workgroup-scan form / the chunk recursion)
from snarkvm_amd import poly
for n_lin in (2049, 100003):
    pa = oracle.fr_op("from_bigint", synthetic.random_fr_integers(n_lin, 4300 + n_lin))
    pz = oracle.fr_op("from_bigint", synthetic.random_fr_integers(1, 4301))
    one = oracle.fr_op("from_bigint", np.array([[1, 0, 0, 0]], dtype=np.uint64))
    q_lin, rem_lin = poly.divide_by_linear(pa, pz)
    wq_lin, _ = oracle.poly_divide(pa, [(0, oracle.fr_op("neg", pz)[0]), (1, one[0])])
    assert np.array_equal(q_lin, wq_lin) and np.array_equal(rem_lin, oracle.poly_evaluate(pa, pz)), ("divide_by_linear", n_lin)
# MSM: a single-round size, a multi-round size over wide windows, a fused batch of proof-sized instances
n = (1 << 19) + 77
bases = oracle.g1_gen_bases(G, 1, n)
sc = synthetic.random_fr_integers(n, 4242)
for tables, bits, m in ((16, 16, 70000), (13, 20, n)):
    rb = msm.RegisteredBases(bases[:m], tables=tables, window_bits=bits)
    assert util.affine_equal(oracle.g1_to_affine(rb.msm(sc[:m])), closed(sc[:m])), ("msm", tables, bits)
    rb.close()
rb = msm.RegisteredBases(bases[:1 << 17], tables=17, window_bits=15)
sizes = [65536, 131072, 40000, 1, 99999]
offs = [0, 0, 1000, 5, 31000]
res = rb.msm_batch([sc[:k] for k in sizes], offsets=offs)
for i, (k, o) in enumerate(zip(sizes, offs)):
    assert util.affine_equal(oracle.g1_to_affine(res[i:i + 1]), closed(sc[:k], start=o + 1)), ("batch", i)
rb.close()
assert util.affine_equal(oracle.g1_to_affine(plugin.msm(bases[:300000], sc[:300000])), closed(sc[:300000])), "ffi"
# G2 (the accumulate arithmetic, tuning lazy2): repeated bases (doublings), registered tables and the one-shot symbol
g2p = synthetic.g2_points(700, distinct=40)
g2s = synthetic.random_fr_integers(700, 777)
want2 = oracle.g2_to_affine(oracle.g2_msm(g2p, g2s)).tobytes()
assert oracle.g2_to_affine(msm.msm_g2(g2p, g2s)).tobytes() == want2, "g2 one-shot"
rg = msm.RegisteredBasesG2(g2p, tables=17, window_bits=15)
assert oracle.g2_to_affine(rg.msm(g2s)).tobytes() == want2, "g2 registered"
rg.close()
print("AB_OK")
'''


@pytest.mark.parametrize("tuning", [
    "ntt_min_tiles=1", "ntt_min_tiles=1024", "lazy=0", "prefetch=0", "acc_one_wg=1,acc_lds=83968", "fuse_batch=0", "fused=0", "seg=96", "reduce_rounds=0", "reduce_rounds=2,seg2=8",
    "hist=1", "ntt_signed=1", "ntt_batch=0", "ring_lanes=5", "coalesce=0", "taper=0", "fuse_max_k=2", "fuse_reduce=0", "lazy2=0", "lazy_tail=1", "xcd=0", "fold_threads2=256", "coalesce_slots=1", "horner2=0", "lazy_tail=0", "fuse_reduce=1", "fuse_reduce=0", "pair2=0"])
def test_ab_switches_are_bit_exact(tuning):
    """csrc/tuning.hip.h: one variable, parsed once per process; every key selects another kernel / launch shape for the same mathematics."""
    r = subprocess.run([sys.executable, "-c", AB_SCRIPT % util.ROOT], capture_output=True, text=True, env=dict(os.environ, SNARKVM_HIP_TUNING=tuning), timeout=900, cwd=util.ROOT)
    assert r.returncode == 0 and "AB_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
