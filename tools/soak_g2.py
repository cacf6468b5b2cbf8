#!/usr/bin/env python3
"""The G2 side of the soak (tools/soak.cpp issues G1 MSMs and Fr transforms only): T Python threads - ctypes releases the GIL inside every call - issue a random mix of
G2 MSM calls over base vectors that REPEAT points (16 / 512 distinct points tiled: equal partial sums meet in the fold / bit-plane trees, the fix kernels of
msm.hip.h run all the time) for a fixed time; every result is compared, after affine normalisation, with the same call issued alone before the threads started.

    python tools/soak_g2.py --seconds 60 --threads 8 --out gpurun_out/r06/soak_g2.json

Families: registered 17 x 15 tables, 2^10 pairs over 16 distinct points (synchronous) | registered 2^14 pairs over 512 distinct points | the host-buffer call
(snarkvm_hip_msm_g2) | a fused batch of three | the call inside an ASYNC_MSM scope beside a G1 MSM of the same thread."""
import argparse, ctypes, json, os, sys, threading, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu as oracle  # noqa: E402  (the checker: affine normalisation only)
from snarkvm_amd import _lib, msm, synthetic  # noqa: E402
from snarkvm_amd.layout import G1_PROJECTIVE, G2_PROJECTIVE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    L = _lib.lib()
    small = synthetic.g2_points(1 << 10, distinct=16)
    big = synthetic.g2_points(1 << 14, distinct=512)
    rg_small = msm.RegisteredBasesG2(small, tables=17, window_bits=15)
    rg_big = msm.RegisteredBasesG2(big, tables=17, window_bits=15)
    from tests import util

    g1 = oracle.g1_gen_bases(util.g1_generator_affine(), 1, 1 << 12)
    rb = msm.RegisteredBases(g1, tables=17, window_bits=15)
    V = 6
    sc_small = [synthetic.random_fr_integers(1 << 10, 5000 + v) for v in range(V)]
    sc_big = [synthetic.random_fr_integers(1 << 14, 6000 + v) for v in range(V)]
    d_small = [torch.from_numpy(x.view(np.int64).reshape(-1).copy()).cuda() for x in sc_small]
    d_g1 = [torch.from_numpy(x[: 1 << 12].view(np.int64).reshape(-1).copy()).cuda() for x in sc_big]
    torch.cuda.synchronize()

    def aff2(p):
        return oracle.g2_to_affine(np.ascontiguousarray(p).view(G2_PROJECTIVE)).tobytes()

    def run(op, v):
        if op == 0:
            return aff2(rg_small.msm(sc_small[v]))
        if op == 1:
            return aff2(rg_big.msm(sc_big[v]))
        if op == 2:
            return aff2(msm.msm_g2(small[: 1000], sc_small[v][: 1000]))
        if op == 3:
            return aff2(rg_small.msm_batch([sc_small[v], sc_small[(v + 1) % V][: 700], sc_small[(v + 2) % V]]))
        out2 = np.zeros(1, dtype=G2_PROJECTIVE)
        out1 = np.zeros(1, dtype=G1_PROJECTIVE)
        _lib.check(L.snarkvm_hip_scope_begin_ex(ctypes.c_void_p(d_small[v].data_ptr()), 3))
        try:
            _lib.check(L.snarkvm_hip_msm_g2_registered(ctypes.c_void_p(out2.ctypes.data), rg_small._h, ctypes.c_size_t(0), ctypes.c_size_t(1 << 10), ctypes.c_void_p(d_small[v].data_ptr()), 1, 0))
            _lib.check(L.snarkvm_hip_msm_registered(ctypes.c_void_p(out1.ctypes.data), rb._h, ctypes.c_size_t(0), ctypes.c_size_t(1 << 12), ctypes.c_void_p(d_g1[v].data_ptr()), 1, 0))
        finally:
            _lib.check(L.snarkvm_hip_scope_end())
        return aff2(out2) + oracle.g1_to_affine(out1).tobytes()

    NOPS = 5
    want = {(op, v): run(op, v) for op in range(NOPS) for v in range(V)}
    again = {(op, v): run(op, v) for op in range(NOPS) for v in range(V)}
    assert want == again, "the solo results are not deterministic after affine normalisation"
    # the small registered call against the oracle's own MSM: the solo results are RIGHT, not merely stable
    for v in range(2):
        assert want[(0, v)] == oracle.g2_to_affine(oracle.g2_msm(small.view(oracle.G2_AFFINE), sc_small[v], oracle.MSM_STANDARD)).tobytes()
    counts = [[0] * NOPS for _ in range(args.threads)]
    problems = []
    stop = time.time() + args.seconds

    def worker(t):
        rng = np.random.default_rng(1000 + t)
        while time.time() < stop and not problems:
            op, v = int(rng.integers(NOPS)), int(rng.integers(V))
            try:
                got = run(op, v)
            except Exception as e:  # noqa: BLE001
                problems.append(f"thread {t} op {op} variant {v}: {e}")
                return
            if got != want[(op, v)]:
                problems.append(f"thread {t} op {op} variant {v}: result differs from the solo call")
                return
            counts[t][op] += 1

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(args.threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    names = ["registered_2^10_16_distinct", "registered_2^14_512_distinct", "host_buffers_1000", "fused_batch_of_three", "async_scope_beside_a_g1_msm"]
    rep = {"ok": not problems, "seconds": args.seconds, "threads": args.threads, "checked_calls": int(sum(map(sum, counts))),
           "calls": {names[i]: int(sum(c[i] for c in counts)) for i in range(NOPS)}, "mismatches_or_errors": problems[:5]}
    print(json.dumps(rep))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rep, open(args.out, "w"), indent=1)
    rg_small.close(); rg_big.close(); rb.close()
    sys.exit(0 if rep["ok"] else 1)


if __name__ == "__main__":
    main()
