#!/usr/bin/env python3
"""BASELINE.json configs[3]: the MSM / NTT / polynomial call pattern of ONE Varuna proof with credits.aleo
transfer_private shapes (|R| = |C| = 2^16, |K| = 2^17) replayed with random data, device resident, timed end
to end on one MI355X - next to the same op list on the CPU oracle (the restated reference functions, 64 threads).  No
circuit or protocol logic: only the hot-path calls, in the order and sizes the prover issues them.

  round 1   iNTT + NTT at |C|, commit w (hiding)
  round 2   3 iNTT at |R|, z_a * z_b on 2|R| (2 NTT + product + iNTT), - z_c, / (X^|R| - 1), commit h_0
  round 3   3 x (iNTT at |C| + product on 2|C|), commit g_1 (hiding), commit h_1
  round 4   3 x (3 iNTT at |K|, one of them coset; one product on 2|K|), commit g_a, g_b, g_c
  round 5   4 commits of the combined polynomials (one pipelined batch), 3 openings (p / (X - z), p(z), MSM)
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import cpu as oracle  # noqa: E402  (CPU baseline leg)
from snarkvm_amd import _lib, synthetic  # noqa: E402
from snarkvm_amd.layout import G1_AFFINE, G1_PROJECTIVE  # noqa: E402

LG_R, LG_K = 16, 17


def main():
    L = _lib.lib()
    torch.cuda.set_device(0)
    nmax = 1 << (LG_K + 1)
    buf = torch.empty((nmax + 8) * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(nmax + 8)))
    h = ctypes.c_void_p()
    _lib.check(L.snarkvm_hip_register_bases_tables(ctypes.byref(h), ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(nmax + 8),
                                                  ctypes.c_size_t(G1_AFFINE.itemsize), 1, 16))
    host = oracle.fr_op("from_bigint", synthetic.random_fr_integers(nmax, 99))
    pool = torch.from_numpy(host.view(np.int64)).cuda()          # source of "random polynomials"
    work = [torch.empty_like(pool) for _ in range(4)]
    point = host[7:8].copy()
    out = np.zeros(1, dtype=G1_PROJECTIVE)
    outs = np.zeros(4, dtype=G1_PROJECTIVE)
    rem = np.zeros((1, 4), dtype=np.uint64)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    acc = {"msm": 0.0, "ntt": 0.0, "poly": 0.0}

    def timed(kind, fn):
        t0 = time.perf_counter()
        fn()
        acc[kind] += time.perf_counter() - t0

    def ntt(t, lg, direction, kind=0):
        timed("ntt", lambda: _lib.check(L.snarkvm_hip_ntt_device(P(t), ctypes.c_uint32(lg), 0, direction, kind)))

    def load(t, n, shift):  # a fresh random vector of n elements (device copy: not part of the hot path)
        t[:n].copy_(pool[shift : shift + n])
        if t.shape[0] > n:
            t[n:].zero_()
        torch.cuda.synchronize()

    def product(a, b, lg):  # PolyMultiplier::multiply of two coefficient vectors on the 2^lg domain, result in a
        ntt(a, lg, 0)
        ntt(b, lg, 0)
        timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_mul_device(P(a), P(a), P(b), ctypes.c_size_t(1 << lg))))
        ntt(a, lg, 1)

    def commit(t, n, hiding=0):
        timed("msm", lambda: _lib.check(L.snarkvm_hip_msm_registered_ex(ctypes.c_void_p(out.ctypes.data), h, 0, n, nmax, hiding, P(t), 1, 1, 0)))

    def proof():
        nR, nK = 1 << LG_R, 1 << LG_K
        a, b, c, d = work
        load(a, nR, 1); ntt(a, LG_R, 1); load(b, nR, 2); ntt(b, LG_R, 0); commit(a, nR - 2, 2)                       # round 1
        for i, t in enumerate((a, b, c)):                                                                       # round 2
            load(t, nR, 10 + i); ntt(t, LG_R, 1)
        d.copy_(c); product(a, b, LG_R + 1)
        timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_vec_op(1, P(a), P(a), P(d), None, None, ctypes.c_size_t(2 * nR), 1)))
        timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_vanishing(P(b), P(c), P(a), ctypes.c_size_t(2 * nR), ctypes.c_size_t(nR), 1)))
        commit(b, nR)
        for m in range(3):                                                                                      # round 3
            load(a, nR, 20 + m); ntt(a, LG_R, 1); load(b, nR, 30 + m); product(a, b, LG_R + 1)
        commit(a, nR - 1, 2); commit(b, nR)
        for m in range(3):                                                                                      # round 4
            load(a, nK, 40 + m); ntt(a, LG_K, 1); load(b, nK, 50 + m); ntt(b, LG_K, 1); load(c, nK, 60 + m); ntt(c, LG_K, 1, 1)
            if m == 0:
                product(a, b, LG_K + 1)
            commit(a, nK - 1)
        ptrs = (ctypes.c_void_p * 4)(*[pool.data_ptr() + 32 * s for s in (3, 5, 9, 11)])                            # round 5
        offs = (ctypes.c_size_t * 4)(0, 0, 0, 0)
        lens = (ctypes.c_size_t * 4)(nK - 2, nK, nR, nK)
        timed("msm", lambda: _lib.check(L.snarkvm_hip_msm_registered_batch(ctypes.c_void_p(outs.ctypes.data), h, 4, offs, lens, ptrs, 1, 1, 0)))
        for s, n in ((13, nK), (17, nR), (19, nK)):                                                             # openings
            load(a, n, s)
            timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_linear(P(b), ctypes.c_void_p(rem.ctypes.data), P(a), ctypes.c_size_t(n),
                                                                                 ctypes.c_void_p(point.ctypes.data), 1)))
            commit(b, n - 1)

    proof()  # warm-up (allocations, twiddle tables)
    for k in acc:
        acc[k] = 0.0
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        proof()
    wall = (time.perf_counter() - t0) / reps
    gpu = {k: v / reps * 1e3 for k, v in acc.items()}
    hot = sum(gpu.values())

    # ---- the same op list on the CPU oracle
    oracle.set_threads(min(64, oracle.max_threads()))
    bases = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=G1_AFFINE)
    cpu = {"msm": 0.0, "ntt": 0.0, "poly": 0.0}

    def ctimed(kind, fn):
        t0 = time.perf_counter()
        r = fn()
        cpu[kind] += time.perf_counter() - t0
        return r

    def cmsm(v):
        ctimed("msm", lambda: oracle.g1_msm(bases[: v.shape[0]], oracle.fr_op("to_bigint", v), oracle.MSM_BATCHED))

    nR, nK = 1 << LG_R, 1 << LG_K
    x = ctimed("ntt", lambda: oracle.ntt(host[:nR], oracle.ORDER_NN, oracle.INVERSE)); ctimed("ntt", lambda: oracle.ntt(host[:nR])); cmsm(x)
    z = [ctimed("ntt", lambda i=i: oracle.ntt(host[i : i + nR], oracle.ORDER_NN, oracle.INVERSE)) for i in range(3)]
    pr = ctimed("ntt", lambda: oracle.polymul(LG_R + 1, [z[0], z[1]]))
    zc = np.zeros_like(pr); zc[:nR] = z[2]
    pr = ctimed("poly", lambda: oracle.fr_vec_op("sub", pr, zc))
    one = oracle.fr_op("from_bigint", np.array([[1, 0, 0, 0]], dtype=np.uint64))
    q, _ = ctimed("poly", lambda: oracle.poly_divide(pr, [(0, oracle.fr_op("neg", one)[0]), (nR, one[0])]))
    cmsm(host[:nR])
    for m in range(3):
        t = ctimed("ntt", lambda: oracle.ntt(host[m : m + nR], oracle.ORDER_NN, oracle.INVERSE))
        ctimed("ntt", lambda: oracle.polymul(LG_R + 1, [t, z[0]]))
    cmsm(host[: nR - 1]); cmsm(host[:nR])
    for m in range(3):
        a_ = ctimed("ntt", lambda: oracle.ntt(host[m : m + nK], oracle.ORDER_NN, oracle.INVERSE))
        b_ = ctimed("ntt", lambda: oracle.ntt(host[m + 5 : m + 5 + nK], oracle.ORDER_NN, oracle.INVERSE))
        ctimed("ntt", lambda: oracle.ntt(host[m + 9 : m + 9 + nK], oracle.ORDER_NN, oracle.INVERSE, oracle.COSET))
        if m == 0:
            ctimed("ntt", lambda: oracle.polymul(LG_K + 1, [a_, b_]))
        cmsm(host[: nK - 1])
    for n in (nK - 2, nK, nR, nK):
        cmsm(host[:n])
    for n in (nK, nR, nK):
        w, _ = ctimed("poly", lambda: oracle.poly_divide(host[:n], [(0, oracle.fr_op("neg", point)[0]), (1, one[0])]))
        cmsm(w)
    chot = sum(cpu.values()) * 1e3
    print("| | MSM (13 commits / openings) ms | NTT (~45 transforms) ms | polynomial passes ms | hot path total ms |")
    print("|---|---|---|---|---|")
    print(f"| 1x MI355X (device resident, synchronous calls; wall {wall * 1e3:.1f} ms incl. the random-vector copies) | {gpu['msm']:.2f} | {gpu['ntt']:.2f} | {gpu['poly']:.2f} | **{hot:.2f}** |")
    print(f"| CPU oracle, {min(64, oracle.max_threads())} threads | {cpu['msm'] * 1e3:.0f} | {cpu['ntt'] * 1e3:.0f} | {cpu['poly'] * 1e3:.0f} | {chot:.0f} |")
    print(f"| ratio | {cpu['msm'] * 1e3 / gpu['msm']:.0f}x | {cpu['ntt'] * 1e3 / gpu['ntt']:.0f}x | {cpu['poly'] * 1e3 / gpu['poly']:.0f}x | {chot / hot:.0f}x |")


if __name__ == "__main__":
    main()
