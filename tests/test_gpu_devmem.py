"""The device-memory calls of the extension ABI (snarkvm_hip_malloc / _free / _memcpy_h2d / _memcpy_d2h / _memcpy_d2d / _memset) and a whole proof's call
list issued WITHOUT torch: every device buffer owned through the C ABI, in a process that never imports torch - what the Rust host of north_star does
through rust/snarkvm-algorithms-hip (resident::DeviceBuffer).  All results against the CPU oracle and against the torch-backed replay."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cpu as oracle
from snarkvm_amd import _lib, proofs, synthetic
from snarkvm_amd.devmem import HipMem
from tests import util
from tests.test_gpu_proofs import _check_against_oracle

pytestmark = pytest.mark.gpu


def test_device_memory_calls_round_trip_and_feed_the_device_entry_points():
    L = _lib.lib()
    n = 1 << 12
    x = oracle.fr_op("from_bigint", synthetic.random_fr_integers(n, 4242))
    a = HipMem.from_numpy(x)
    assert np.array_equal(a.download(dtype=np.uint64).reshape(-1, 4), x)
    # a transform on memory the library allocated, outside any scope
    _lib.check(L.snarkvm_hip_ntt_device(ctypes.c_void_p(a.ptr), ctypes.c_uint32(12), 0, 0, 0))
    want = oracle.ntt(x)
    assert np.array_equal(a.download(dtype=np.uint64).reshape(-1, 4), want)
    # d2d + memset outside a scope are complete on return
    b = HipMem(2 * a.nbytes)
    b.fill(0, 0xFF, b.nbytes)
    b.copy_from(a.nbytes, a.ptr, a.nbytes)
    b.fill(0, 0, 64)
    got = b.download()
    assert not got[:64].any() and (got[64 : a.nbytes] == 0xFF).all() and np.array_equal(got[a.nbytes :].view(np.uint64).reshape(-1, 4), want)
    # partial copies at an offset
    part = x[100:200]
    a.upload(part, byte_offset=32 * 7)
    assert np.array_equal(a.download(32 * 100, 32 * 7, np.uint64).reshape(-1, 4), part)
    # inside a scope d2d / memset are enqueued in call order with the transforms; the host-side copies wait for the scope's stream
    a.upload(x)
    _lib.check(L.snarkvm_hip_scope_begin(ctypes.c_void_p(a.ptr)))
    try:
        b.fill(0, 0, b.nbytes)
        b.copy_from(0, a.ptr, a.nbytes)            # b[:n] = x
        _lib.check(L.snarkvm_hip_ntt_device(ctypes.c_void_p(b.ptr), ctypes.c_uint32(13), 0, 0, 0))   # zero-padded to 2^13
        mid = b.download(dtype=np.uint64).reshape(-1, 4)  # d2h inside the scope: ordered behind the queued transform, complete on return
        _lib.check(L.snarkvm_hip_ntt_device(ctypes.c_void_p(b.ptr), ctypes.c_uint32(13), 0, 1, 0))
    finally:
        _lib.check(L.snarkvm_hip_scope_end())
    padded = np.concatenate([x, np.zeros_like(x)])
    assert np.array_equal(mid, oracle.ntt(padded))
    assert np.array_equal(b.download(dtype=np.uint64).reshape(-1, 4), padded)
    a.free()
    b.free()
    a.free()  # idempotent


def test_device_memory_argument_errors_are_reported_not_fatal():
    L = _lib.lib()
    p = ctypes.c_void_p(1)
    _lib.check(L.snarkvm_hip_malloc(ctypes.byref(p), 0, -1))
    assert not p.value  # zero bytes: a null block, free(NULL) is fine
    _lib.check(L.snarkvm_hip_free(None))
    with pytest.raises(_lib.HipError):
        _lib.check(L.snarkvm_hip_malloc(ctypes.byref(p), 64, 99))  # no such logical device
    with pytest.raises(_lib.HipError):
        _lib.check(L.snarkvm_hip_malloc(None, 64, -1))
    m = HipMem(4096)
    with pytest.raises(_lib.HipError):
        m.copy_from(0, m.at(100), 1000)  # overlapping ranges
    host = np.zeros(16, dtype=np.uint8)
    with pytest.raises(_lib.HipError):
        _lib.check(L.snarkvm_hip_memcpy_h2d(host.ctypes.data, host.ctypes.data, 16))  # destination is not device memory
    with pytest.raises(_lib.HipError):
        _lib.check(L.snarkvm_hip_memcpy_d2d(m.ptr, host.ctypes.data, 16))  # source is not device memory
    _lib.check(L.snarkvm_hip_memcpy_h2d(m.ptr, None, 0))  # empty copies are no-ops
    # in-place device division is refused (header: quotient must not overlap poly)
    z = np.zeros((1, 4), dtype=np.uint64)
    with pytest.raises(_lib.HipError):
        _lib.check(L.snarkvm_hip_fr_divide_by_linear(ctypes.c_void_p(m.ptr), ctypes.c_void_p(z.ctypes.data), ctypes.c_void_p(m.ptr), ctypes.c_size_t(64), ctypes.c_void_p(z.ctypes.data), 1))
    m.free()


def test_hipmem_replay_equals_torch_replay_in_one_process():
    """The same keys seed, the same salts: the proof replayed on HipMem buffers and on torch tensors gives the same 15 group elements."""
    shape = proofs.ProofShape(lg_r=12, lg_k=13, lg_g2=10)
    k_hip = proofs.ProverKeys(shape, seed=21, mem="hip")
    k_torch = proofs.ProverKeys(shape, seed=21)
    assert np.array_equal(k_hip.g1_host.view(np.uint8), k_torch.g1_host.view(np.uint8))
    w_hip, w_torch = proofs.SingleProofWorkspace(k_hip), proofs.SingleProofWorkspace(k_torch)
    for salt in (0, 3):
        a, b = [], []
        proofs.replay_single(w_hip, salt, a, async_msm=True, await_rounds=True, msm_in_stream=True)
        proofs.replay_single(w_torch, salt, b, async_msm=True, await_rounds=True, msm_in_stream=True)
        assert proofs.normalize_results(a) == proofs.normalize_results(b)
    k_hip.close()
    k_torch.close()


def test_torch_free_process_replays_a_proof_every_result_vs_oracle(tmp_path):
    """A process that never imports torch (SNARKVM_HIP_NO_TORCH=1; the helper asserts 'torch' not in sys.modules at its end) allocates every buffer through
    snarkvm_hip_malloc, uploads the pool with snarkvm_hip_memcpy_h2d, produces the operands with snarkvm_hip_memcpy_d2d / _memset inside the scope and
    replays the proof in four modes (enqueued, synchronous, awaited, awaited in-stream): all 15 results of every mode and proof against the oracle's
    restatement of the data flow; a warmed replay grows no library workspace."""
    out = tmp_path / "torch_free.npz"
    env = dict(os.environ, SNARKVM_HIP_NO_TORCH="1")
    salts = [0, 5]
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "helpers", "torch_free_replay.py"), str(out), "8", ",".join(map(str, salts))],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "OK torch-free replay" in r.stdout, r.stdout + r.stderr
    res = np.load(out)
    shape = proofs.ProofShape(lg_r=12, lg_k=13, lg_g2=10)
    keys = proofs.ProverKeys(shape, seed=8)  # the same seed: the same pool, bases and points (host copies are what the oracle needs)
    for salt in salts:
        ref = None
        for mode in ("async", "sync", "await", "in_stream"):
            g1 = res[f"{mode}_{salt}_g1"].tobytes()
            got = [g1[144 * j : 144 * (j + 1)] for j in range(14)] + [res[f"{mode}_{salt}_g2"].tobytes()]
            if mode == "async":
                _check_against_oracle(keys, shape, salt, got)
                ref = proofs.normalize_results(got)
            else:
                assert proofs.normalize_results(got) == ref, (salt, mode)
    # (several logical devices: device pointers are dealt round-robin over the logical devices of their GPU, a later replay may meet a lane for the first time -
    # the same reason test_second_batch_of_a_shape_allocates_nothing_and_is_not_slower skips there)
    if int(res["num_devices"][0]) == 1:
        assert not res["alloc_stats"][:4].any(), res["alloc_stats"]
    keys.close()
