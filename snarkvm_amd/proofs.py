"""The hot-path call pattern of one Varuna proof (BASELINE.json configs[3] / [4]) replayed on the gfx950 backend.

No circuit and no protocol logic: only the MSM / NTT / polynomial calls the prover issues, in the order and sizes of SURVEY.md
3.1 for `credits.aleo/transfer_private` (|R| = |C| = 2^16, |K| = 2^17), on device-resident random data:

  round 1   iNTT + NTT at |C|, commit w (hiding)                                          first.rs:127-160
  round 2   3 iNTT at |R|, z_a * z_b on 2|R| (2 NTT + product + iNTT), - z_c, / (X^|R| - 1), commit h_0   second.rs:104-170
  round 3   3 x (iNTT at |C| + product on 2|C|), commit g_1 (hiding), commit h_1           third.rs:158-317
  round 4   3 x (3 iNTT at |K|, one of them coset; one product on 2|K|), commit g_a, g_b, g_c   fourth.rs:174-231
  round 5   4 commits of the combined polynomials (one pipelined batch), 3 openings (p / (X - z), p(z), MSM)
            fifth.rs:50-66, sonic_pc/mod.rs:316-337
  + one G2 MSM of 2^16 pairs per proof (north_star's G2 leg; the reference prover itself issues none)

`ProofBatch` replays many such proofs concurrently: worker threads (the reference's rayon workers, one commitment / proof
each) call the C ABI at the same time and the backend hands every call its own (device, stream) lane; with several devices
in use the proofs' working sets are spread over them and every call runs where its data lives.  Their proof-sized MSMs meet in
the library's coalescer (csrc/runtime.hip.h::msm_coalesced) and travel as fused groups.

`LockstepBatch` replays P proofs in LOCK STEP from one thread - `VarunaSNARK::prove_batch` (snark/varuna/varuna.rs:336) is a batch
by construction: step k of all P proofs is issued together, i.e. round k's commitments of all proofs are ONE
snarkvm_hip_msm_registered_batch_ex call (P x m instances -> fused groups), the transforms of a step are ONE
snarkvm_hip_ntt_device_batch call per size (one kernel launch per pass for up to 48 vectors), the pointwise passes are ONE strided
launch sequence per step (snarkvm_hip_fr_*_strided), and none of these waits for the GPU by itself (snarkvm_hip_scope_begin / _end).
"""
import ctypes
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib, synthetic
from .layout import G1_AFFINE, G1_PROJECTIVE, G2_AFFINE, G2_PROJECTIVE


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class ProofShape:
    """Sizes of one proof's domains; defaults = transfer_private (test_credits.rs:2868-2897)."""

    def __init__(self, lg_r=16, lg_k=17, lg_g2=16):
        self.lg_r, self.lg_k, self.lg_g2 = lg_r, lg_k, lg_g2
        self.nmax = 1 << (lg_k + 1)

    def pairs(self):
        """scalar-point pairs of the G1 commitments / openings of one proof"""
        nR, nK = 1 << self.lg_r, 1 << self.lg_k
        return (nR) + nR + (nR + 1) + nR + 3 * (nK - 1) + (nK - 2 + nK + nR + nK) + (nK - 1 + nR - 1 + nK - 1)


class ProverKeys:
    """The static device-resident operands shared by every proof: registered G1 powers (+ the gamma powers behind them) with
    17 precomputed tables of 15-bit windows (the default; `tables` x `window_bits` must cover 254 bits), a registered G2 vector (same geometry), and a pool of random Fr data the proofs slice their "polynomials" from.
    Registration replicates the bases to every device the backend uses."""

    def __init__(self, shape, seed=99, tables=17, window_bits=15, mem="torch", scalars="uniform"):
        """scalars: the distribution of the pool the proofs slice their polynomials from - "uniform" Fr values, or "witness" (50 % zero,
        25 % below 2^16, 25 % uniform, as VALUES: the pool holds their Montgomery images).  Only the commitments that read the pool directly (round 5: 4 of the
        14) see that distribution - the others commit to outputs of transforms and divisions, which are uniform whatever went in.
        mem: who owns the device buffers of the keys and of the workspaces built on them - "torch" (tensors) or "hip" (snarkvm_amd.devmem.HipMem:
        snarkvm_hip_malloc / _memcpy_* through the C ABI; no torch anywhere on the path - what a Rust host does)."""
        assert mem in ("torch", "hip")
        self.mem = mem
        self.shape = shape
        self.geometry = (tables, window_bits)
        L = _lib.lib()
        n = shape.nmax + 8
        if mem == "hip":
            from .devmem import HipMem

            buf = HipMem(n * G1_AFFINE.itemsize)
        else:
            import torch

            buf = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
        _lib.check(L.snarkvm_hip_g1_generate_bases_device(_p(buf), ctypes.c_uint64(1), ctypes.c_size_t(n)))
        self.h = ctypes.c_void_p()
        _lib.check(L.snarkvm_hip_register_bases_windowed(ctypes.byref(self.h), _p(buf), ctypes.c_size_t(n), ctypes.c_size_t(G1_AFFINE.itemsize), 1, tables, window_bits))
        self.g1_host = (buf.download() if mem == "hip" else buf.cpu().numpy()).view(G1_AFFINE)
        if mem == "hip":
            buf.free()
        del buf
        # G2: the generator's multiples would need Fq2 point generation on the host; a G2 vector of repeated (decoded) real
        # points with random scalars exercises the same kernels
        self.hg2 = ctypes.c_void_p()
        self.g2_host = None
        if shape.lg_g2:
            self.g2_host = synthetic.g2_points(1 << shape.lg_g2)
            _lib.check(L.snarkvm_hip_register_bases_g2(ctypes.byref(self.hg2), ctypes.c_void_p(self.g2_host.ctypes.data), ctypes.c_size_t(self.g2_host.shape[0]),
                                                       ctypes.c_size_t(G2_AFFINE.itemsize), tables, window_bits))
        assert scalars in ("uniform", "witness")
        self.scalars = scalars
        if scalars == "witness":
            self.pool_host = synthetic.witness_like_fr_montgomery(shape.nmax + 4096, seed)
        else:
            self.pool_host = synthetic.random_fr_integers(shape.nmax + 4096, seed)  # any residue < r is a valid Montgomery image
        self.point = self.pool_host[7:8].copy()

    def close(self):
        L = _lib.lib()
        if self.h:
            L.snarkvm_hip_free_bases(self.h)
            self.h = ctypes.c_void_p()
        if self.hg2:
            L.snarkvm_hip_free_bases_g2(self.hg2)
            self.hg2 = ctypes.c_void_p()


class ProofWorkspace:
    """Device buffers of one in-flight proof on one device (four work vectors of the largest domain + the data pool)."""

    def __init__(self, keys, device_index=0):
        import torch

        self.keys = keys
        self.device = torch.device("cuda", device_index)
        with torch.cuda.device(self.device):
            self.pool = torch.from_numpy(keys.pool_host.view(np.int64).reshape(-1)).to(self.device)
            self.work = [torch.empty(keys.shape.nmax * 4, dtype=torch.int64, device=self.device) for _ in range(4)]
            torch.cuda.synchronize()
        self.out = np.zeros(1, dtype=G1_PROJECTIVE)
        self.outs = np.zeros(4, dtype=G1_PROJECTIVE)
        self.out_g2 = np.zeros(1, dtype=G2_PROJECTIVE)
        self.rem = np.zeros((1, 4), dtype=np.uint64)
        self.times = {"msm": 0.0, "ntt": 0.0, "poly": 0.0, "g2": 0.0}


def replay(ws, salt=0, collect=None):
    """Issue the hot-path calls of one proof on workspace `ws`.  `salt` shifts the slices of the random pool the proof reads, so
    different proofs commit to different polynomials.  collect: a list that receives a copy of every commitment (144 B each;
    the G2 result last) for checking."""
    import torch

    L = _lib.lib()
    keys = ws.keys
    sh = keys.shape
    nR, nK = 1 << sh.lg_r, 1 << sh.lg_k
    a, b, c, d = ws.work
    pool = ws.pool
    t = ws.times

    def timed(kind, fn):
        t0 = time.perf_counter()
        fn()
        t[kind] += time.perf_counter() - t0

    def ntt(v, lg, direction, kind=0):
        timed("ntt", lambda: _lib.check(L.snarkvm_hip_ntt_device(_p(v), ctypes.c_uint32(lg), 0, direction, kind)))

    def ntt_batch(vs, lg, directions):  # independent transforms of one round: one enqueue, one synchronisation
        k = len(vs)
        ptrs = (ctypes.c_void_p * k)(*[v.data_ptr() for v in vs])
        dirs = (ctypes.c_int * k)(*directions)
        timed("ntt", lambda: _lib.check(L.snarkvm_hip_ntt_device_batch(ptrs, ctypes.c_size_t(k), ctypes.c_uint32(lg), 0, dirs, None)))

    def load(v, n, shift):  # a fresh "polynomial" of n coefficients (device copy on torch's stream: not part of the hot path)
        s = 4 * (shift + salt)
        with torch.cuda.device(ws.device):
            v[: 4 * n].copy_(pool[s : s + 4 * n])
            if v.shape[0] > 4 * n:
                v[4 * n :].zero_()
            torch.cuda.current_stream().synchronize()

    def product(x, y, lg):  # PolyMultiplier::multiply of two coefficient vectors on the 2^lg domain, result in x
        ntt_batch((x, y), lg, (0, 0))
        timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_mul_device(_p(x), _p(x), _p(y), ctypes.c_size_t(1 << lg))))
        ntt(x, lg, 1)

    def commit_round(polys):
        """SonicKZG10::commit of one round (sonic_pc/mod.rs:177-257): every (vector, length, hiding degree) of the round in ONE
        batched call - plaintext MSM over powers[0 .. n) + hiding MSM over the gamma powers each, Fr::to_bigint fused."""
        k = len(polys)
        ptrs = (ctypes.c_void_p * k)(*[v.data_ptr() if hasattr(v, "data_ptr") else int(v) for v, _, _ in polys])
        off0 = (ctypes.c_size_t * k)(*([0] * k))
        n0 = (ctypes.c_size_t * k)(*[n for _, n, _ in polys])
        off1 = (ctypes.c_size_t * k)(*([sh.nmax] * k))
        n1 = (ctypes.c_size_t * k)(*[h for _, _, h in polys])
        outs = np.zeros(k, dtype=G1_PROJECTIVE)
        timed("msm", lambda: _lib.check(L.snarkvm_hip_msm_registered_batch_ex(ctypes.c_void_p(outs.ctypes.data), keys.h, k, off0, n0, off1, n1, ptrs, 1, 1, 0)))
        if collect is not None:
            collect.extend(outs[i : i + 1].tobytes() for i in range(k))

    load(a, nR, 1); ntt(a, sh.lg_r, 1); load(b, nR, 2); ntt(b, sh.lg_r, 0); commit_round([(a, nR - 2, 2)])         # round 1
    for i, v in enumerate((a, b, c)):                                                                             # round 2
        load(v, nR, 10 + i)
    ntt_batch((a, b, c), sh.lg_r, (1, 1, 1))  # z_a, z_b, z_c (second.rs:104-113)
    with torch.cuda.device(ws.device):
        d.copy_(c)
        torch.cuda.current_stream().synchronize()
    product(a, b, sh.lg_r + 1)
    timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_vec_op(1, _p(a), _p(a), _p(d), None, None, ctypes.c_size_t(2 * nR), 1)))
    timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_vanishing(_p(b), _p(c), _p(a), ctypes.c_size_t(2 * nR), ctypes.c_size_t(nR), 1)))
    commit_round([(b, nR, 0)])
    for m in range(3):                                                                                            # round 3
        load(a, nR, 20 + m); ntt(a, sh.lg_r, 1); load(b, nR, 30 + m); product(a, b, sh.lg_r + 1)
    commit_round([(a, nR - 1, 2), (b, nR, 0)])                                                                    # g_1 (hiding), h_1
    r4 = (a, b, c)                                                                                                # round 4: g_a, g_b, g_c
    for m in range(3):
        v = r4[m]
        load(v, nK, 40 + m); ntt(v, sh.lg_k, 1); load(d, nK, 50 + m); ntt(d, sh.lg_k, 1); load(d, nK, 60 + m); ntt(d, sh.lg_k, 1, 1)
        if m == 0:
            load(d, nK, 70); product(v, d, sh.lg_k + 1)
    commit_round([(a, nK - 1, 0), (b, nK - 1, 0), (c, nK - 1, 0)])
    base = pool.data_ptr()                                                                                        # round 5
    commit_round([(base + 32 * (3 + salt), nK - 2, 0), (base + 32 * (5 + salt), nK, 0), (base + 32 * (9 + salt), nR, 0), (base + 32 * (11 + salt), nK, 0)])
    opens = []                                                                                                    # openings
    for (s, n), q in zip(((13, nK), (17, nR), (19, nK)), (b, c, d)):
        load(a, n, s)
        timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_linear(_p(q), ctypes.c_void_p(ws.rem.ctypes.data), _p(a), ctypes.c_size_t(n),
                                                                            ctypes.c_void_p(keys.point.ctypes.data), 1)))
        opens.append((q, n - 1, 0))
    commit_round(opens)                                                                                           # batch_open: the three witness commitments
    if keys.hg2:                                                                                                 # G2 leg
        n2 = 1 << sh.lg_g2
        timed("g2", lambda: _lib.check(L.snarkvm_hip_msm_g2_registered(ctypes.c_void_p(ws.out_g2.ctypes.data), keys.hg2, 0, n2,
                                                                        ctypes.c_void_p(pool.data_ptr() + 32 * (23 + salt)), 1, 0)))
        if collect is not None:
            collect.append(ws.out_g2.tobytes())


def normalize_results(results):
    """Commitment lists as comparable values: the device returns Jacobian representatives whose coordinates depend on the
    order in which bucket entries happened to be added, so results are compared after affine normalisation (as the reference's
    own tests do, variable_base/mod.rs:96-105).  G1 records (144 B) go through snarkvm_hip_g1_to_affine; the G2 record (288 B)
    is normalised with Python integers."""
    from . import kzg10

    out = []
    for item in results:
        if len(item) == 144:
            out.append(kzg10.to_affine(np.frombuffer(item, dtype=G1_PROJECTIVE)).tobytes())
        else:
            q = synthetic.Q_MOD
            rinv = pow(1 << 384, q - 2, q)
            w = np.frombuffer(item, dtype="<u8").reshape(6, 6)
            c = [sum(int(l) << (64 * i) for i, l in enumerate(row)) * rinv % q for row in w]  # X.c0, X.c1, Y.c0, Y.c1, Z.c0, Z.c1
            X, Y, Z = (c[0], c[1]), (c[2], c[3]), (c[4], c[5])
            if Z == (0, 0):
                out.append(b"infinity")
                continue
            zi = synthetic._fq2_inv(Z)
            zi2 = synthetic._fq2_mul(zi, zi)
            out.append(repr((synthetic._fq2_mul(X, zi2), synthetic._fq2_mul(Y, synthetic._fq2_mul(zi2, zi)))).encode())
    return out


class SingleProofWorkspace:
    """Device buffers of ONE proof proved at a time by one caller thread (BASELINE.json configs[3]): every vector of the proof has
    its own row, so that the independent transforms of a round can travel as one batch and nothing has to wait for a buffer."""

    ROWS = 28

    def __init__(self, keys, device_index=0, mem=None):
        """mem (default: the keys' kind): "torch" - pool and work matrix are tensors, operand copies are strided tensor copies on the scope's stream;
        "hip" - both are HipMem blocks (snarkvm_hip_malloc), operand copies are snarkvm_hip_memcpy_d2d / _memset calls, which a scope enqueues on its
        stream like any other device-resident call.  The replayed call list and its 15 results are the same."""
        self.keys = keys
        self.mem = mem or keys.mem
        self.row_bytes = keys.shape.nmax * 32
        if self.mem == "hip":
            from .devmem import HipMem

            self.device = None
            # (snarkvm_hip_malloc counts LOGICAL devices - entries of the library's device list; a process that drives one GPU, whatever its CUDA index, has logical device 0)
            device_index = device_index if device_index < _lib.lib().snarkvm_hip_num_devices() else 0
            self.pool = HipMem.from_numpy(keys.pool_host, device_index)
            self.work = HipMem(self.ROWS * self.row_bytes, device_index)
            self.work.fill(0, 0, self.work.nbytes)
        else:
            import torch

            self.device = torch.device("cuda", device_index)
            with torch.cuda.device(self.device):
                self.pool = torch.from_numpy(keys.pool_host.view(np.int64).reshape(-1)).to(self.device)
                self.work = torch.zeros((self.ROWS, keys.shape.nmax * 4), dtype=torch.int64, device=self.device)
                torch.cuda.synchronize()
        self.work_ptr = self.work.data_ptr()
        self._stream = None
        self.outs = np.zeros(14, dtype=G1_PROJECTIVE)
        self.out_g2 = np.zeros(1, dtype=G2_PROJECTIVE)
        self.rem = np.zeros((3, 4), dtype=np.uint64)
        self.times = {"enqueue": 0.0, "wait": 0.0}

    def bind_stream(self, hip_stream):
        """the scope's stream, for the operand copies of the torch kind (the hip kind's copies are calls of the library: ordered by the scope itself)"""
        if self.mem == "torch":
            import torch

            self._stream = torch.cuda.ExternalStream(hip_stream, device=self.device)

    def load(self, r, n, shift, count=1, zero_to=0):
        """rows r .. r + count - 1 <- pool[shift + i ...] (n coefficients each; row i starts one element later), zeros up to `zero_to`; enqueued on
        the scope's stream - "the prover produced a polynomial" """
        if self.mem == "hip":
            for i in range(count):
                self.work.copy_from((r + i) * self.row_bytes, self.pool.at(32 * (shift + i)), 32 * n)
                if zero_to > n:
                    self.work.fill((r + i) * self.row_bytes + 32 * n, 0, 32 * (zero_to - n))
            return
        import torch

        w, pool = self.work, self.pool
        with torch.cuda.stream(self._stream):
            w[r : r + count, : 4 * n].copy_(pool.as_strided((count, 4 * n), (4, 1), 4 * shift))
            if zero_to > n:
                w[r : r + count, 4 * n : 4 * zero_to].zero_()


def _clocks():
    return (time.clock_gettime_ns(time.CLOCK_MONOTONIC), time.clock_gettime_ns(time.CLOCK_MONOTONIC_RAW), time.clock_gettime_ns(time.CLOCK_BOOTTIME), time.time_ns())


def replay_single(ws, salt=0, collect=None, async_msm=True, marks=None, await_rounds=False, msm_in_stream=False, g2_after=0):
    """The hot-path calls of ONE proof from ONE caller thread, issued for latency (configs[3]; the reference proves one transaction at a
    time: synthesizer/snark/src/proving_key/mod.rs:37 -> VarunaSNARK::prove_batch, varuna.rs:336).  Same calls, sizes and operands as
    `replay` - the 15 results are the same group elements - but:
      * the whole proof is ONE deferred-synchronisation scope (snarkvm_hip_scope_begin_ex): no call waits for the GPU, the operand
        copies ("the prover produced a polynomial") go onto the scope's own stream (snarkvm_hip_scope_stream), and with
        SNARKVM_HIP_SCOPE_ASYNC_MSM the commitments of round k are only enqueued - they run on further streams beside the transforms of
        round k + 1, their host Horner finishes run in snarkvm_hip_scope_end while the GPU works on the later rounds;
      * the independent transforms of a round - the three matrices of rounds 3 and 4, which the reference hands to a job pool
        (third.rs:160-175, fourth.rs:174-190) - are ONE batched call per size instead of one call per vector;
      * the independent G2 MSM is issued first and runs underneath everything else.
    async_msm=False: the same call list with synchronous MSMs (the A/B of the overlap).
    await_rounds=True (with async_msm): after every commitment round the host WAITS for that round's results (snarkvm_hip_scope_collect) before it
    issues the next round - the order a real prover is bound to, because the commitments of round k enter the Fiat-Shamir transcript that yields
    the challenge of round k + 1 (this replay takes its challenges as inputs, so nothing here needs them); the scope's stream, the earlier tails
    and the G2 MSM keep running meanwhile.
    msm_in_stream=True (with await_rounds): once the G2 MSM is out on its own stream, the scope is switched to SNARKVM_HIP_SCOPE_MSM_IN_STREAM - the
    awaited commitment rounds run on the scope's own stream, in order with the transforms, without an event hand-off between streams per round.
    marks: a list that receives (label, clocks) after every step was issued (tools/proof1_timeline.py lines them up with a kernel trace).
    The workspace decides who owns the device memory (SingleProofWorkspace: torch tensors, or HipMem blocks through the C ABI - no torch on the path)."""

    def mark(label):
        if marks is not None:
            marks.append((label, _clocks()))

    mark("begin")

    L = _lib.lib()
    keys = ws.keys
    sh = keys.shape
    nR, nK = 1 << sh.lg_r, 1 << sh.lg_k
    pool = ws.pool
    stride_b = ws.row_bytes
    estride = ctypes.c_size_t(ws.row_bytes // 32)
    t0 = time.perf_counter()

    def row(r):
        return ws.work_ptr + r * stride_b

    def rp(r):
        return ctypes.c_void_p(row(r))

    # every committed vector keeps its row until the proof is done (SNARKVM_HIP_SCOPE_STABLE_INPUTS: the transform stream never waits for an MSM)
    _lib.check(L.snarkvm_hip_scope_begin_ex(ctypes.c_void_p(pool.data_ptr()), 3 if async_msm else 0))
    try:
        ws.bind_stream(L.snarkvm_hip_scope_stream())

        def load(r, n, shift, count=1, zero_to=0):
            ws.load(r, n, shift + salt, count, zero_to)

        def ntt(rows, lg, direction, kind=0):
            k = len(rows)
            ptrs = (ctypes.c_void_p * k)(*[row(r) for r in rows])
            dirs = (ctypes.c_int * k)(*([direction] * k))
            kinds = (ctypes.c_int * k)(*([kind] * k))
            _lib.check(L.snarkvm_hip_ntt_device_batch(ptrs, ctypes.c_size_t(k), ctypes.c_uint32(lg), 0, dirs, kinds))

        def mul(x, y, lg, count=1):  # rows x .. x + count - 1 *= rows y .. y + count - 1
            _lib.check(L.snarkvm_hip_fr_vec_op_strided(2, rp(x), rp(x), rp(y), None, None, ctypes.c_size_t(1 << lg), ctypes.c_size_t(count), estride))

        slot = [0]

        def commit_round(polys):
            """(pointer, n, hiding) per commitment; results land in ws.outs[slot ...] when the scope ends"""
            k = len(polys)
            ptrs = (ctypes.c_void_p * k)(*[p for p, _, _ in polys])
            off0 = (ctypes.c_size_t * k)(*([0] * k))
            n0 = (ctypes.c_size_t * k)(*[n for _, n, _ in polys])
            off1 = (ctypes.c_size_t * k)(*([sh.nmax] * k))
            n1 = (ctypes.c_size_t * k)(*[h for _, _, h in polys])
            out = ctypes.c_void_p(ws.outs.ctypes.data + G1_PROJECTIVE.itemsize * slot[0])
            _lib.check(L.snarkvm_hip_msm_registered_batch_ex(out, keys.h, k, off0, n0, off1, n1, ptrs, 1, 1, 0))
            slot[0] += k
            if await_rounds and async_msm:
                _lib.check(L.snarkvm_hip_scope_collect(out))  # this round's commitments; the G2 MSM and nothing else is waited for
            rounds_done[0] += 1
            if rounds_done[0] == g2_after:
                issue_g2()

        in_stream = msm_in_stream and await_rounds and async_msm

        def issue_g2():                                                                      # G2 leg: independent of everything else, on a further stream
            if keys.hg2:
                if in_stream:
                    _lib.check(L.snarkvm_hip_scope_set_flags(3))
                _lib.check(L.snarkvm_hip_msm_g2_registered(ctypes.c_void_p(ws.out_g2.ctypes.data), keys.hg2, 0, 1 << sh.lg_g2,
                                                            ctypes.c_void_p(pool.data_ptr() + 32 * (23 + salt)), 1, 0))
            mark("g2 msm issued")
            if in_stream:
                _lib.check(L.snarkvm_hip_scope_set_flags(3 | 4))

        # g2_after = k: the independent MSM is issued behind the k-th commitment round (0: first of all, the default; measured, bench.py --workload proof1: issuing it
        # later does not help a proof in transcript order - profiles/r06_summary.md)
        rounds_done = [0]
        if g2_after <= 0:
            issue_g2()
        load(26, nR, 1, count=2)                                                             # round 1: rows 26, 27
        ntt([26], sh.lg_r, 1); ntt([27], sh.lg_r, 0)
        mark("round 1 transforms issued")
        commit_round([(row(26), nR - 2, 2)])
        mark("round 1 commit issued")
        load(0, nR, 10, count=3, zero_to=2 * nR)                                             # round 2: z_a, z_b, z_c in rows 0, 1, 2
        ntt([0, 1, 2], sh.lg_r, 1)
        ntt([0, 1], sh.lg_r + 1, 0); mul(0, 1, sh.lg_r + 1); ntt([0], sh.lg_r + 1, 1)
        _lib.check(L.snarkvm_hip_fr_vec_op(1, rp(0), rp(0), rp(2), None, None, ctypes.c_size_t(2 * nR), 1))
        _lib.check(L.snarkvm_hip_fr_divide_by_vanishing(rp(1), rp(3), rp(0), ctypes.c_size_t(2 * nR), ctypes.c_size_t(nR), 1))
        mark("round 2 transforms + passes issued")
        commit_round([(row(1), nR, 0)])
        mark("round 2 commit issued")
        load(4, nR, 20, count=3, zero_to=2 * nR); load(7, nR, 30, count=3, zero_to=2 * nR)   # round 3: a_m in rows 4-6, b_m in rows 7-9
        ntt([4, 5, 6], sh.lg_r, 1)
        ntt([4, 7, 5, 8, 6, 9], sh.lg_r + 1, 0); mul(4, 7, sh.lg_r + 1, count=3); ntt([4, 5, 6], sh.lg_r + 1, 1)
        mark("round 3 transforms issued")
        commit_round([(row(6), nR - 1, 2), (row(9), nR, 0)])                                 # g_1 (hiding), h_1
        mark("round 3 commits issued")
        load(10, nK, 40, count=3, zero_to=2 * nK); load(13, nK, 50, count=3); load(16, nK, 60, count=3); load(19, nK, 70, zero_to=2 * nK)   # round 4
        ntt([10, 11, 12, 13, 14, 15], sh.lg_k, 1)
        ntt([16, 17, 18], sh.lg_k, 1, 1)
        ntt([10, 19], sh.lg_k + 1, 0); mul(10, 19, sh.lg_k + 1); ntt([10], sh.lg_k + 1, 1)
        mark("round 4 transforms issued")
        commit_round([(row(10), nK - 1, 0), (row(11), nK - 1, 0), (row(12), nK - 1, 0)])
        mark("round 4 commits issued")
        base = pool.data_ptr()                                                               # round 5: straight from the pool
        commit_round([(base + 32 * (o + salt), n, 0) for o, n in ((3, nK - 2), (5, nK), (9, nR), (11, nK))])
        mark("round 5 commits issued")
        opens = []                                                                           # openings
        for i, (s, n) in enumerate(((13, nK), (17, nR), (19, nK))):
            load(20 + i, n, s)
            _lib.check(L.snarkvm_hip_fr_divide_by_linear(rp(23 + i), ctypes.c_void_p(ws.rem[i : i + 1].ctypes.data), rp(20 + i), ctypes.c_size_t(n),
                                                          ctypes.c_void_p(keys.point.ctypes.data), 1))
            opens.append((row(23 + i), n - 1, 0))
        mark("opening divisions issued")
        commit_round(opens)
        mark("opening commits issued")
        t1 = time.perf_counter()
    finally:
        end = L.snarkvm_hip_scope_end()
    _lib.check(end)
    mark("scope_end returned (all 15 results on the host)")
    t2 = time.perf_counter()
    ws.times["enqueue"] += t1 - t0
    ws.times["wait"] += t2 - t1
    if collect is not None:
        collect.extend(ws.outs[i : i + 1].tobytes() for i in range(14))
        if keys.hg2:
            collect.append(ws.out_g2.tobytes())


class LockstepWorkspace:
    """Device buffers of P proofs replayed in lock step on one device: four work matrices [P, nmax] (row p = proof p) + the data pool."""

    def __init__(self, keys, count, device_index=0):
        import torch

        self.keys = keys
        self.count = count
        self.device = torch.device("cuda", device_index)
        with torch.cuda.device(self.device):
            self.pool = torch.from_numpy(keys.pool_host.view(np.int64).reshape(-1)).to(self.device)
            self.work = [torch.empty((count, keys.shape.nmax * 4), dtype=torch.int64, device=self.device) for _ in range(4)]
            torch.cuda.synchronize()
        self.rem = np.zeros((3, count, 4), dtype=np.uint64)
        self.times = {"msm": 0.0, "ntt": 0.0, "poly": 0.0, "g2": 0.0, "load": 0.0, "wait": 0.0}


def replay_lockstep(ws, salts, collect=False, async_scope=False, await_rounds=False):
    """The hot-path calls of len(salts) <= ws.count proofs, issued step by step for all proofs together (same calls, sizes and
    operands as `replay` per proof: results are the same group elements).  Returns per proof the list of its 14 commitments (+ the
    G2 result) when collect is set.
    async_scope=True (round 5, measured and NOT the default): the whole group is ONE SNARKVM_HIP_SCOPE_ASYNC_MSM scope - the operand copies go
    onto the scope's stream, a round's fused MSM call is only enqueued (on one of the scope's further streams; the work matrices are reused
    by the next steps, which wait on the GPU until the MSM has read them) and finished by scope_end, the G2 batch runs underneath.  The
    fused groups of 32 - 128 instances fill the chip by themselves, so running them beside each other buys nothing and the extra streams
    cost: 188 - 191 proofs/s against 199 - 207 for the default form (a scope per step, synchronous commitment calls; profiles/r05_summary.md)."""
    import torch

    L = _lib.lib()
    keys = ws.keys
    sh = keys.shape
    P = len(salts)
    assert 0 < P <= ws.count
    nR, nK = 1 << sh.lg_r, 1 << sh.lg_k
    A, B, C, D = range(4)
    pool = ws.pool
    t = ws.times
    stride = ws.work[0].stride(0) * 8
    results = [[] for _ in range(P)] if collect else None
    step = salts[1] - salts[0] if P > 1 else 1
    arithmetic = all(salts[i] == salts[0] + i * step for i in range(P)) and step > 0

    def vec(v, p):
        return ws.work[v].data_ptr() + p * stride

    stream = None
    deferred = []  # (proof-major outputs array, instances per proof) of the commitment calls, read after scope_end
    if async_scope:
        _lib.check(L.snarkvm_hip_scope_begin_ex(ctypes.c_void_p(pool.data_ptr()), 1))
        stream = torch.cuda.ExternalStream(L.snarkvm_hip_scope_stream(), device=ws.device)

    class scope:  # device-resident calls between loads: enqueued on one stream, one wait at the end (async_scope: the group's one scope is open already)
        def __enter__(self):
            if not async_scope:
                _lib.check(L.snarkvm_hip_scope_begin(ctypes.c_void_p(pool.data_ptr())))

        def __exit__(self, *exc):
            if not async_scope:
                _lib.check(L.snarkvm_hip_scope_end())
            return False

    def timed(kind, fn):
        t0 = time.perf_counter()
        fn()
        t[kind] += time.perf_counter() - t0

    def load(v, n, shift):  # a fresh "polynomial" of n coefficients per proof (device copies on torch's stream: not part of the hot path)
        t0 = time.perf_counter()
        w = ws.work[v]
        with torch.cuda.device(ws.device), torch.cuda.stream(stream):
            if arithmetic:
                w[:P, : 4 * n].copy_(pool.as_strided((P, 4 * n), (4 * step, 1), 4 * (shift + salts[0])))
            else:
                for p in range(P):
                    s0 = 4 * (shift + salts[p])
                    w[p, : 4 * n].copy_(pool[s0 : s0 + 4 * n])
            if w.shape[1] > 4 * n:
                w[:P, 4 * n :].zero_()
            if not async_scope:
                torch.cuda.current_stream().synchronize()
        t["load"] += time.perf_counter() - t0

    def ntt_all(vs, lg, direction, kind=0):  # the same transform of vectors `vs` of every proof: one call, one launch per pass per 48 vectors
        ptr_list = [vec(v, p) for v in vs for p in range(P)]
        k = len(ptr_list)
        ptrs = (ctypes.c_void_p * k)(*ptr_list)
        dirs = (ctypes.c_int * k)(*([direction] * k))
        kinds = (ctypes.c_int * k)(*([kind] * k))
        timed("ntt", lambda: _lib.check(L.snarkvm_hip_ntt_device_batch(ptrs, ctypes.c_size_t(k), ctypes.c_uint32(lg), 0, dirs, kinds)))

    estride = ctypes.c_size_t(stride // 32)  # elements between the vectors of consecutive proofs

    def vp(v):
        return ctypes.c_void_p(vec(v, 0))

    def product(x, y, lg):  # PolyMultiplier::multiply per proof, result in x
        ntt_all((x, y), lg, 0)
        timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_vec_op_strided(2, vp(x), vp(x), vp(y), None, None, ctypes.c_size_t(1 << lg), ctypes.c_size_t(P), estride)))
        ntt_all((x,), lg, 1)

    def commit_round(polys):
        """One SonicKZG10::commit round of ALL proofs: P x len(polys) instances in one batched call (proof-major).  polys: (pointer(p), n, hiding)."""
        m = len(polys)
        k = m * P
        ptrs = (ctypes.c_void_p * k)(*[fp(p) for p in range(P) for fp, _, _ in polys])
        off0 = (ctypes.c_size_t * k)(*([0] * k))
        n0 = (ctypes.c_size_t * k)(*[n for _ in range(P) for _, n, _ in polys])
        off1 = (ctypes.c_size_t * k)(*([sh.nmax] * k))
        n1 = (ctypes.c_size_t * k)(*[h for _ in range(P) for _, _, h in polys])
        outs = np.zeros(k, dtype=G1_PROJECTIVE)
        timed("msm", lambda: _lib.check(L.snarkvm_hip_msm_registered_batch_ex(ctypes.c_void_p(outs.ctypes.data), keys.h, k, off0, n0, off1, n1, ptrs, 1, 1, 0)))
        if async_scope and await_rounds:  # this round's commitments now (the transcript order); the G2 batch and the scope's stream keep running
            timed("msm", lambda: _lib.check(L.snarkvm_hip_scope_collect(ctypes.c_void_p(outs.ctypes.data))))
        deferred.append((outs, m))  # (asynchronous scope: written by scope_end / scope_collect)

    def work_ptr(v):
        return lambda p: vec(v, p)

    load(A, nR, 1); load(B, nR, 2)                                                                                 # round 1
    with scope():
        ntt_all((A,), sh.lg_r, 1); ntt_all((B,), sh.lg_r, 0)
    commit_round([(work_ptr(A), nR - 2, 2)])
    for i, v in enumerate((A, B, C)):                                                                             # round 2
        load(v, nR, 10 + i)
    with scope():
        ntt_all((A, B, C), sh.lg_r, 1)
    with torch.cuda.device(ws.device), torch.cuda.stream(stream):
        ws.work[D][:P].copy_(ws.work[C][:P])
        if not async_scope:
            torch.cuda.current_stream().synchronize()
    with scope():
        product(A, B, sh.lg_r + 1)
        timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_vec_op_strided(1, vp(A), vp(A), vp(D), None, None, ctypes.c_size_t(2 * nR), ctypes.c_size_t(P), estride)))
        timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_vanishing_strided(vp(B), vp(C), vp(A), ctypes.c_size_t(2 * nR), ctypes.c_size_t(nR), ctypes.c_size_t(P), estride)))
    commit_round([(work_ptr(B), nR, 0)])
    for m in range(3):                                                                                            # round 3
        load(A, nR, 20 + m); load(B, nR, 30 + m)
        with scope():
            ntt_all((A,), sh.lg_r, 1)
            product(A, B, sh.lg_r + 1)
    commit_round([(work_ptr(A), nR - 1, 2), (work_ptr(B), nR, 0)])                                                # g_1 (hiding), h_1
    r4 = (A, B, C)                                                                                                # round 4: g_a, g_b, g_c
    for m in range(3):
        v = r4[m]
        load(v, nK, 40 + m); load(D, nK, 50 + m)
        with scope():
            ntt_all((v, D), sh.lg_k, 1)
        load(D, nK, 60 + m)
        with scope():
            ntt_all((D,), sh.lg_k, 1, 1)
        if m == 0:
            load(D, nK, 70)
            with scope():
                product(v, D, sh.lg_k + 1)
    commit_round([(work_ptr(A), nK - 1, 0), (work_ptr(B), nK - 1, 0), (work_ptr(C), nK - 1, 0)])
    base = pool.data_ptr()                                                                                        # round 5
    commit_round([((lambda p, o=o: base + 32 * (o + salts[p])), n, 0) for o, n in ((3, nK - 2), (5, nK), (9, nR), (11, nK))])
    opens = []                                                                                                    # openings
    for i, ((s, n), q) in enumerate(zip(((13, nK), (17, nR), (19, nK)), (B, C, D))):
        load(A, n, s)
        with scope():
            timed("poly", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_linear_strided(vp(q), ctypes.c_void_p(ws.rem[i].ctypes.data), vp(A), ctypes.c_size_t(n),
                                                                                        ctypes.c_void_p(keys.point.ctypes.data), ctypes.c_size_t(P), estride)))
        opens.append((work_ptr(q), n - 1, 0))
    commit_round(opens)                                                                                           # batch_open: the three witness commitments
    if keys.hg2:                                                                                                 # G2 leg: one batched call
        n2 = 1 << sh.lg_g2
        ptrs = (ctypes.c_void_p * P)(*[pool.data_ptr() + 32 * (23 + salts[p]) for p in range(P)])
        offs = (ctypes.c_size_t * P)(*([0] * P))
        ns = (ctypes.c_size_t * P)(*([n2] * P))
        outs2 = np.zeros(P, dtype=G2_PROJECTIVE)
        timed("g2", lambda: _lib.check(L.snarkvm_hip_msm_g2_registered_batch(ctypes.c_void_p(outs2.ctypes.data), keys.hg2, P, offs, ns, ptrs, 1, 0)))
    else:
        outs2 = None
    if async_scope:
        timed("wait", lambda: _lib.check(L.snarkvm_hip_scope_end()))
    if collect:
        for outs, m in deferred:
            for p in range(P):
                results[p].extend(outs[p * m + j : p * m + j + 1].tobytes() for j in range(m))
        if outs2 is not None:
            for p in range(P):
                results[p].append(outs2[p : p + 1].tobytes())
    return results


class LockstepBatch:
    """`count` proofs in groups of `group` replayed in lock step by one thread per device (see the module docstring)."""

    def __init__(self, keys, group=16, devices=None, async_scope=False, await_rounds=False):
        import torch

        self.keys = keys
        self.async_scope = async_scope
        self.await_rounds = await_rounds
        ndev = torch.cuda.device_count()
        devices = list(range(ndev)) if devices is None else list(devices)
        self.group = group
        self.workspaces = [LockstepWorkspace(keys, group, d) for d in devices]

    def run(self, salts, collect=False):
        """Replay one proof per entry of `salts`; returns (wall seconds, [commitment lists] or None)."""
        nws = len(self.workspaces)
        chunks = [salts[i : i + self.group] for i in range(0, len(salts), self.group)]
        results = [None] * len(chunks)

        def one(ci):
            try:
                results[ci] = replay_lockstep(self.workspaces[ci % nws], chunks[ci], collect, self.async_scope, self.await_rounds)
            except BaseException:
                err = _lib.lib().snarkvm_hip_scope_end()  # a failed call must not leave this thread's scope open
                if err.message:
                    _lib._libc.free(err.message)
                raise

        t0 = time.perf_counter()
        if nws == 1:
            for ci in range(len(chunks)):
                one(ci)
        else:  # one thread per device, each taking every nws-th group
            def dev_worker(w):
                for ci in range(w, len(chunks), nws):
                    one(ci)
            with ThreadPoolExecutor(nws) as ex:
                list(ex.map(dev_worker, range(nws)))
        dt = time.perf_counter() - t0
        return dt, ([r for ch in results for r in ch] if collect else None)


class ProofBatch:
    """`count` proofs replayed by `workers` concurrent caller threads (BASELINE.json configs[4]: 64 proofs; one process per GPU
    takes its share, or one process drives every device the backend uses)."""

    def __init__(self, keys, workers=4, devices=None, scope=False, async_msm=True):
        """scope=True: every worker issues its proof inside ONE asynchronous scope (`replay_single`: nothing waits for the GPU until the
        proof's results are due, a round's independent transforms are one batched call); False: one synchronous call per step (`replay`)."""
        import torch

        self.keys = keys
        self.scope = scope
        self.async_msm = async_msm  # scope mode only; False: the commitment rounds are synchronous calls (and meet other callers' in the coalescer)
        ndev = torch.cuda.device_count()
        devices = list(range(ndev)) if devices is None else list(devices)
        cls = SingleProofWorkspace if scope else ProofWorkspace
        self.workspaces = [cls(keys, devices[w % len(devices)]) for w in range(workers)]
        self._free = list(self.workspaces)
        self._lock = threading.Lock()

    def run(self, salts, collect=False):
        """Replay one proof per entry of `salts`; returns (wall seconds, [commitment lists] or None)."""
        results = [None] * len(salts)

        def one(i):
            with self._lock:
                ws = self._free.pop()
            try:
                got = [] if collect else None
                if self.scope:
                    replay_single(ws, salts[i], got, self.async_msm)
                else:
                    replay(ws, salts[i], got)
                results[i] = got
            finally:
                with self._lock:
                    self._free.append(ws)

        t0 = time.perf_counter()
        if len(self.workspaces) == 1:
            for i in range(len(salts)):
                one(i)
        else:
            with ThreadPoolExecutor(len(self.workspaces)) as ex:
                list(ex.map(one, range(len(salts))))
        return time.perf_counter() - t0, (results if collect else None)


# ---- the same proof through the reference's OWN three symbols (what an unmodified snarkVM gets) ---------------------------------------------------
class FfiProofHost:
    """Host-side operands of the drop-in replay: every vector is caller-owned HOST memory, like the `Vec<Fr>` / `&[G1Affine]` / `&[BigInteger256]` a Rust
    caller passes.  `bases` is ONE long-lived vector - `powers_of_beta_g` followed by the gamma powers - and every commitment passes a slice
    of it at the offsets KZG10 uses (kzg10/mod.rs:117-119), which is what the opt-in base cache keys on."""

    def __init__(self, keys, threads=4):
        self.keys = keys
        self.bases = np.ascontiguousarray(keys.g1_host)  # stays alive and in place for the life of this object
        self.pool = np.ascontiguousarray(keys.pool_host, dtype=np.uint64).reshape(-1, 4)
        self.threads = threads
        self.ex = ThreadPoolExecutor(threads)
        self._conv = None

    def close(self):
        self.ex.shutdown()
        if self._conv is not None:
            self._conv.free()
            self._conv = None

    def to_bigint(self, v):
        """`convert_to_bigints` (kzg10/mod.rs:469-474): the reference runs it on the CPU between the transform and the MSM.  Glue, not one of the three
        symbols: done here through device memory of the library (HipMem + snarkvm_hip_fr_convert_device) and never inside a timed step."""
        from .devmem import HipMem

        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        if self._conv is None or self._conv.nbytes < v.nbytes:
            if self._conv is not None:
                self._conv.free()
            self._conv = HipMem(max(v.nbytes, 32 << 18))
        self._conv.upload(v)
        _lib.check(_lib.lib().snarkvm_hip_fr_convert_device(ctypes.c_void_p(self._conv.ptr), ctypes.c_void_p(self._conv.ptr), ctypes.c_size_t(v.shape[0]), 1))
        return self._conv.download(v.nbytes, 0, np.uint64).reshape(-1, 4)


def replay_ffi(host, salt=0, collect=None, times=None, g2=True):
    """The data flow of `replay` - the same 15 results - with every hot-path step issued through the three symbols `snarkvm-algorithms-cuda` binds
    (algorithms/cuda/src/lib.rs:42-69) on HOST buffers: each transform one `snarkvm_ntt` (fft/domain.rs:375-391), each product one `snarkvm_polymul`
    (fft/polynomial/multiplier.rs:79-95), each commitment one `snarkvm_msm` over a slice of the long-lived base vector (msm/variable_base/mod.rs:33-43),
    the commitments of a round - and the per-matrix work of rounds 3 and 4 - issued from `host.threads` caller threads like the reference's rayon workers
    (polycommit/sonic_pc/mod.rs:186-245; third.rs:160-175, fourth.rs:174-190).  What the reference does on the CPU between those calls - convert_to_bigints,
    the pointwise subtraction, the two polynomial divisions, the <= 3-point hiding MSM (below the plugin's `len > 1024` gate, variable_base/mod.rs:32-35) and
    the final mixed addition - is GLUE: run outside the timed steps (through host-operand calls of this library, any correct implementation would do).
    times: dict that receives seconds per step class {"ntt", "polymul", "msm", "g2", "glue"} plus call counts; the sum of ntt + polymul + msm is the time one
    proof spends inside the accelerator boundary of an unmodified snarkVM.  g2: additionally one `snarkvm_hip_msm_g2` over host buffers (an extension - the
    reference never sends a G2 MSM to its plugin), accounted separately."""
    L = _lib.lib()
    keys = host.keys
    sh = keys.shape
    nR, nK, nmax = 1 << sh.lg_r, 1 << sh.lg_k, sh.nmax
    pool, bases = host.pool, host.bases
    t = times if times is not None else {}
    for k in ("ntt", "polymul", "msm", "g2", "glue"):
        t.setdefault(k, 0.0)
    for k in ("ntt_calls", "polymul_calls", "msm_calls"):
        t.setdefault(k, 0)
    P = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731

    def load(n, shift, size=None):
        v = np.zeros((size or n, 4), dtype=np.uint64)
        v[:n] = pool[shift + salt : shift + salt + n]
        return v

    def ntt(v, direction, kind=0):  # in place, the whole array is the domain
        lg = v.shape[0].bit_length() - 1
        assert v.shape[0] == 1 << lg
        _lib.check(L.snarkvm_ntt(P(v), ctypes.c_uint32(lg), 0, direction, kind))
        return v

    def polymul(x, y, lg):  # PolyMultiplier::multiply of two coefficient vectors -> a new vector of the whole 2^lg domain
        out = np.zeros((1 << lg, 4), dtype=np.uint64)
        ptrs = (ctypes.c_void_p * 2)(x.ctypes.data, y.ctypes.data)
        lens = (ctypes.c_size_t * 2)(x.shape[0], y.shape[0])
        _lib.check(L.snarkvm_polymul(P(out), ctypes.c_size_t(2), ptrs, lens, ctypes.c_size_t(0), None, None, ctypes.c_uint32(lg)))
        return out

    def msm(off, n, scalars):  # one commitment's plaintext MSM: a slice of the long-lived base vector + canonical scalars
        out = np.zeros(1, dtype=G1_PROJECTIVE)
        _lib.check(L.snarkvm_msm(P(out), ctypes.c_void_p(bases.ctypes.data + off * G1_AFFINE.itemsize), ctypes.c_size_t(n), P(scalars), ctypes.c_size_t(G1_AFFINE.itemsize)))
        return out

    def step(kind, jobs, ncalls):
        """one timed step: `jobs` (callables) issued concurrently from the caller threads; returns their results in order"""
        t0 = time.perf_counter()
        res = [jobs[0]()] if len(jobs) == 1 else list(host.ex.map(lambda f: f(), jobs))
        t[kind] += time.perf_counter() - t0
        return res

    def glue(fn):
        t0 = time.perf_counter()
        r = fn()
        t["glue"] += time.perf_counter() - t0
        return r

    commits = []  # (plain part, [(hiding offset, scalars)] or None) in result order

    def commit_round(polys):
        """polys: (vector, n, hiding).  Glue: convert_to_bigints; timed: one snarkvm_msm per polynomial from the caller threads."""
        scs = glue(lambda: [host.to_bigint(v[: n + h]) for v, n, h in polys])
        res = step("msm", [lambda sc=sc, n=n: msm(0, n, np.ascontiguousarray(sc[:n])) for sc, (_, n, _) in zip(scs, polys)], len(polys))
        t["msm_calls"] += len(polys)
        for r, sc, (_, n, h) in zip(res, scs, polys):
            commits.append((r, np.ascontiguousarray(sc[n : n + h]) if h else None))

    # round 1 (first.rs:127-160)
    a, b = load(nR, 1), load(nR, 2)
    step("ntt", [lambda: ntt(a, 1)], 1); step("ntt", [lambda: ntt(b, 0)], 1)
    t["ntt_calls"] += 2
    commit_round([(a, nR - 2, 2)])
    # round 2 (second.rs:104-170)
    z = [load(nR, 10 + i) for i in range(3)]
    step("ntt", [lambda v=v: ntt(v, 1) for v in z], 3)
    t["ntt_calls"] += 3
    prod = step("polymul", [lambda: polymul(z[0], z[1], sh.lg_r + 1)], 1)[0]
    t["polymul_calls"] += 1

    def rowcheck():
        zc = np.zeros_like(prod)
        zc[:nR] = z[2]
        row = np.empty_like(prod)
        _lib.check(L.snarkvm_hip_fr_vec_op(1, P(row), P(prod), P(zc), None, None, ctypes.c_size_t(2 * nR), 0))
        q, rem = np.zeros((nR, 4), dtype=np.uint64), np.zeros((nR, 4), dtype=np.uint64)
        _lib.check(L.snarkvm_hip_fr_divide_by_vanishing(P(q), P(rem), P(row), ctypes.c_size_t(2 * nR), ctypes.c_size_t(nR), 0))
        return q

    q = glue(rowcheck)
    commit_round([(q, nR, 0)])
    # round 3 (third.rs:158-317): the three matrices on three caller threads
    def matrix3(m):
        tm = ntt(load(nR, 20 + m), 1)
        b_in = load(nR, 30 + m)
        a_vec = polymul(tm, b_in, sh.lg_r + 1)
        b_vec = ntt(load(nR, 30 + m, 2 * nR), 0)  # the replay leaves `b` holding its forward transform on the doubled domain
        return a_vec, b_vec

    r3 = step("ntt", [lambda m=m: matrix3(m) for m in range(3)], 9)  # (accounted under "ntt": 2 transforms + 1 product per matrix)
    t["ntt_calls"] += 6
    t["polymul_calls"] += 3
    a_vec, b_vec = r3[2]
    commit_round([(a_vec, nR - 1, 2), (b_vec, nR, 0)])
    # round 4 (fourth.rs:174-231)
    def matrix4(m):
        v = ntt(load(nK, 40 + m), 1)
        ntt(load(nK, 50 + m), 1)
        ntt(load(nK, 60 + m), 1, 1)
        if m == 0:
            v = polymul(v, load(nK, 70), sh.lg_k + 1)
        return v

    r4 = step("ntt", [lambda m=m: matrix4(m) for m in range(3)], 10)
    t["ntt_calls"] += 9
    t["polymul_calls"] += 1
    commit_round([(v, nK - 1, 0) for v in r4])
    # round 5 (fifth.rs:50-66): straight from the pool
    commit_round([(pool[o + salt : o + salt + n], n, 0) for o, n in ((3, nK - 2), (5, nK), (9, nR), (11, nK))])
    # openings (sonic_pc/mod.rs:316-337, kzg10/mod.rs:213-236)
    def witness(s, n):
        p = load(n, s)
        qq, rem = np.zeros((n - 1, 4), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64)
        _lib.check(L.snarkvm_hip_fr_divide_by_linear(P(qq), P(rem), P(p), ctypes.c_size_t(n), P(keys.point), 0))
        return qq

    ws_ = glue(lambda: [witness(s, n) for s, n in ((13, nK), (17, nR), (19, nK))])
    commit_round([(w, w.shape[0], 0) for w in ws_])
    out_g2 = None
    if g2 and keys.g2_host is not None and sh.lg_g2:
        n2 = 1 << sh.lg_g2
        sc2 = np.ascontiguousarray(pool[23 + salt : 23 + salt + n2])
        out_g2 = np.zeros(1, dtype=G2_PROJECTIVE)
        t0 = time.perf_counter()
        _lib.check(L.snarkvm_hip_msm_g2(P(out_g2), P(keys.g2_host), ctypes.c_size_t(n2), P(sc2), ctypes.c_size_t(G2_AFFINE.itemsize)))
        t["g2"] += time.perf_counter() - t0
    if collect is not None:
        def finish():  # the hiding term (<= 3 points: the reference's CPU path) and the closing addition
            for plain, hid in commits:
                if hid is None:
                    collect.append(plain.tobytes())
                    continue
                both = np.zeros(2, dtype=G1_PROJECTIVE)
                both[0] = plain[0]
                both[1] = msm(nmax, hid.shape[0], hid)[0]
                tot = np.zeros(1, dtype=G1_PROJECTIVE)
                _lib.check(L.snarkvm_hip_g1_sum(P(tot), P(both), ctypes.c_size_t(2)))
                collect.append(tot.tobytes())
            if out_g2 is not None:
                collect.append(out_g2.tobytes())

        glue(finish)
    return t
