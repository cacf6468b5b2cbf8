#!/usr/bin/env python3
"""bench.py - BLS12-377 G1 MSM (pairs/s) + Fr NTT (elements/s) on MI355X.

Default workload (`--workload msm`): one "step" = one G1 variable-base MSM of 2^lg_msm scalar-point pairs (BASELINE.json
configs[1]: 2^24) with bases registered in HBM and scalars resident in HBM when the timed region starts.  `value` = pairs/s
over all ranks (each rank runs its own independent MSM instances: instance-level sharding, no data-path collective -> weak
scaling).  Every timed result is checked outside the timed region (closed form sum_i s_i (i+1) G, evaluated by a one-point MSM
on the device; on rank 0 at N = 1 also against the CPU oracle, together with the full 2^24 NTT output).  The JSON line also
carries: the NTT throughput at 2^24, the 2^20 half of BASELINE.json's metric, the throughput without precomputed tables, the
registration cost, the end-to-end times of the reference's own FFI symbols over host buffers, `roofline` objects following
SURVEY.md 8(d) (algorithmic bytes n * 128 + 144 for the whole MSM, 32 n for the scalar-read phase, 64 n for an NTT), an
`alu_roofline` against the wall-clock-calibrated arithmetic ceilings of tools/ecbench.hip, and a `cpu_baseline` (the C++
restatement of the reference's rayon path, oracle/, timed on this box's host cores on a bounded sample).

`--workload proof1`: BASELINE.json configs[3] - ONE Varuna-proof-shaped call list at a time from one caller thread (the reference
proves one transaction at a time, proving_key/mod.rs:37): `value` = proofs/s of 32 proofs proved one after the other,
`ms_per_step` = the latency of one proof; every result of every timed proof is compared with the serial replay and all 15 results
of two proofs with the CPU oracle.  The default workload carries compact `proof1`, `proofs64` and `concurrent_callers` legs too.

`--workload proofs64`: BASELINE.json configs[4] - 64 Varuna-proof-shaped call lists (14 G1 commitments / openings of
2^16-2^17, ~45 NTTs, the polynomial passes, one 2^16 G2 MSM each; snarkvm_amd/proofs.py), 64 / N proofs per rank, replayed in
lock step (`value`, proofs/s over all ranks, strong scaling) and by concurrent caller threads (`concurrent_callers`); one whole
proof is checked against the CPU oracle.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime maps streams onto 4 hardware queues by default; the lanes of concurrent callers (proofs64: 8 caller threads
# per rank) want one each.  Read once at HIP initialisation, so it has to be in the environment before torch touches the GPU
# (measured on proofs64, 8 callers: 97.7 -> 104 proofs/s).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

DTYPE = "u32 limbs (29-bit radix) modular integer arithmetic, Fq 377-bit / Fr 253-bit"
G1_GEN_X = [1171681672315280277, 6528257384425852712, 7514971432460253787, 2032708395764262463, 12876543207309632302, 107509843840671767]
G1_GEN_Y = [13572190014569192121, 15344828677741220784, 17067903700058808083, 10342263224753415805, 1083990386877464092, 21335464879237822]
R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041


def duplicate_devices(rank_devices):
    """UUIDs that more than one rank reports (two ranks sharing a GPU make `value` meaningless as a scaling point): [] when every rank has its own device."""
    seen = {}
    for d in rank_devices or []:
        u = d.get("uuid") if isinstance(d, dict) else None
        if u:
            seen.setdefault(u, []).append(d.get("rank"))
    return [{"uuid": u, "ranks": r} for u, r in seen.items() if len(r) > 1]


def weighted_sum_mod_r(scalars, start=1):
    """sum_i (start + i) * scalars[i] mod r for an (n,4) u64 array, vectorised (16-bit pieces, chunked)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    total = 0
    CH = 1 << 18
    for lo in range(0, s.shape[0], CH):
        hi = min(s.shape[0], lo + CH)
        wgt = np.arange(start + lo, start + hi, dtype=np.uint64)
        for limb in range(4):
            col = s[lo:hi, limb]
            for piece in range(4):
                part = (col >> np.uint64(16 * piece)) & np.uint64(0xFFFF)
                total += int(np.sum(part * wgt, dtype=np.uint64)) << (64 * limb + 16 * piece)  # weights < 2^26, pieces < 2^16, chunk 2^18
    return total % R_MOD


class SclkSampler:
    """Samples the GPU's shader clock (amdgpu hwmon freq1_input, Hz) from a host thread while a timed region runs: the effective clock
    the round-3 review asked for next to `mad_frac` (the arithmetic ceilings were calibrated at the 2.4 GHz peak clock; dense VALU work
    sustains less).  A box exposes the hwmon files of every card of its node, visible to this process or not, in no useful order: all of
    them are sampled and the one that ran fastest is reported (the GPU under load); None when nothing readable ran above 500 MHz."""

    def __init__(self, dev_index=0, period_s=0.002):
        import glob
        import threading

        self.paths = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        self.period = period_s
        self.samples = {p: [] for p in self.paths}
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if self.paths else None

    def _run(self):
        live = list(self.paths)
        while live and not self._stop.is_set():
            for p in list(live):
                try:
                    with open(p) as f:
                        self.samples[p].append(int(f.read().strip()) / 1e6)
                except Exception:
                    live.remove(p)
            self._stop.wait(self.period)

    def __enter__(self):
        if self._th:
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th:
            self._th.join(timeout=1.0)
        return False

    def summary(self):
        best = None
        for p, v in self.samples.items():
            if v and (best is None or sum(v) / len(v) > best[0]):
                best = (sum(v) / len(v), p, sorted(v))
        if best is None or best[0] < 500.0:
            return None
        mean, p, v = best
        return {"mean_mhz": mean, "min_mhz": v[0], "max_mhz": v[-1], "median_mhz": v[len(v) // 2], "samples": len(v), "source": p,
                "cards_sampled": len(self.paths)}


def load_profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["msm", "proofs64", "proof1"], default="msm")
    ap.add_argument("--lg-msm", type=int, default=24)
    ap.add_argument("--lg-ntt", type=int, default=24)
    ap.add_argument("--ntt-steps", type=int, default=10)
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--tables", type=int, default=0, help="precomputed 2^(table_bits*j) multiples of the registered bases (0: from --table-bits)")
    ap.add_argument("--table-bits", type=int, default=-1,
                    help="bits per base table = widest bucket window; -1: 22 / 20 / 16 by size; tables = ceil(254 / table_bits)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="issue the K steps one synchronous MSM at a time instead of one pipelined batch of K independent MSMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the 2^20 / tables1 / FFI legs (they are outside the timed region anyway)")
    ap.add_argument("--cpu-lg-msm", type=int, default=24, help="CPU baseline sample: 2^k pairs of the same workload (24 = the full configuration, ~10 s on 64 threads)")
    ap.add_argument("--cpu-lg-ntt", type=int, default=24)
    ap.add_argument("--proofs", type=int, default=64)
    ap.add_argument("--proof-workers", type=int, default=8, help="concurrent caller threads per rank (proofs64, callers mode)")
    ap.add_argument("--proof-geometry", default="17x15", help="base tables x window bits of the proofs' registered SRS (proofs64)")
    ap.add_argument("--proof-group", type=int, default=32, help="proofs replayed in lock step per group (proofs64, lockstep mode)")
    ap.add_argument("--lockstep-async", action="store_true", help="proofs64: the whole lock-step group in ONE asynchronous scope (measured slower: 191 vs 203 proofs/s)")
    ap.add_argument("--proof1-sync-msm", action="store_true", help="proof1: synchronous commitments (the A/B of SNARKVM_HIP_SCOPE_ASYNC_MSM)")
    ap.add_argument("--no-proof-legs", action="store_true", help="default workload: skip the proof1 / proofs64 / concurrent_callers legs")
    ap.add_argument("--scalars", choices=["uniform", "witness"], default="uniform",
                    help="distribution of the timed legs' scalars: uniform in [0, r), or witness-like (SURVEY.md 8d config 2: 50 %% zero, 25 %% < 2^16, 25 %% uniform; "
                         "synthesizer/process/src/tests/test_credits.rs:2868-2925 shapes).  The default line reports BOTH for the headline, proof1 and proofs64 "
                         "(value_witness_like, ...); this flag makes the second distribution the headline itself")
    ap.add_argument("--proof-mem", choices=["torch", "hip"], default="hip",
                    help="proof1: who owns the proof's device buffers and issues its operand copies - \"hip\": snarkvm_hip_malloc / _memcpy_d2d / _memset through the C ABI (what a "
                         "Rust host does; no torch on the data path), \"torch\": tensors and strided tensor copies on the scope's stream (rounds 4 - 5)")
    ap.add_argument("--ffi-only", action="store_true", help="proof1: the same proof through the reference's OWN three symbols on host buffers (what an unmodified snarkVM gets): "
                                                             "rows stateless / SNARKVM_HIP_BASE_CACHE=16 / resident")
    ap.add_argument("--ffi-threads", type=int, default=4, help="--ffi-only: caller threads that issue the commitments of a round (the reference's rayon workers)")
    ap.add_argument("--proof1-all-at-once", action="store_true", help="proof1: make the everything-enqueued-at-once replay (an order no Fiat-Shamir prover can use) the headline again")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; a launcher that already narrowed HIP_VISIBLE_DEVICES to one device per rank leaves index 0
    dev_index = local_rank if (world > 1 and torch.cuda.device_count() > local_rank) else 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        backend = os.environ.get("SNARKVM_BENCH_BACKEND", "nccl")  # "gloo": smoke-testing the N > 1 path on a 1-GPU box
        # A rank that cannot bring up RCCL (no xGMI peer access, IPC mode, a dead GPU) must END the job with a message, not leave the
        # other ranks waiting in their first collective: bounded initialisation, then one probe all-reduce under the same bound.
        import datetime

        limit = datetime.timedelta(seconds=int(os.environ.get("SNARKVM_BENCH_INIT_TIMEOUT_S", "180")))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=limit)
            else:
                dist.init_process_group(backend, timeout=limit)
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"probe all-reduce returned {probe.item()} instead of {world}")
        except Exception as e:  # noqa: BLE001
            print(f"[bench rank {rank}] FATAL: cannot initialise the {backend} process group on cuda:{dev_index} within {limit.seconds} s: {e!r}", file=sys.stderr, flush=True)
            os._exit(3)  # no destructors: a half-initialised communicator may block in its own teardown
    else:
        torch.cuda.set_device(0)

    def device_identity():
        """what this rank runs on: lets a scaling run be checked for two ranks sharing one GPU"""
        pr = torch.cuda.get_device_properties(dev_index)
        return {"rank": rank, "cuda_index": dev_index, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")), "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}

    def gather_objects(obj):
        if world > 1:
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        return [obj]

    from snarkvm_amd import _lib, plugin, synthetic
    from snarkvm_amd.layout import G1_AFFINE, G1_PROJECTIVE
    from snarkvm_amd.msm import RegisteredBases

    L = _lib.lib()
    _lib.check(L.snarkvm_hip_set_device(ctypes.c_int(dev_index)))  # this process drives exactly one GPU

    def barrier():
        torch.cuda.synchronize()
        _lib.check(L.snarkvm_hip_synchronize())
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    def gather_over_ranks(dt):
        """every rank's own time of a region (rank order): lets a scaling run be checked rank by rank against N x the single-GPU figure"""
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            out = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            return [float(x.item()) for x in out]
        return [dt]

    rank_devices = gather_objects(device_identity())
    if args.workload == "proofs64":
        return proofs64(args, rank, world, dev_index, barrier, max_over_ranks, gather_over_ranks, rank_devices)
    if args.workload == "proof1":
        return proof1(args, rank, world, dev_index, barrier, max_over_ranks, gather_over_ranks, rank_devices)

    gen = np.zeros(1, dtype=G1_AFFINE)
    gen["x"] = G1_GEN_X
    gen["y"] = G1_GEN_Y

    def device_multiple_of_g(k):
        """k * G as an affine record, by a one-point MSM on the device (the plain FFI path with its own planner)."""
        sc = np.array([[(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64)
        return to_affine(plugin.msm(gen, sc))

    def to_affine(proj):
        proj = np.ascontiguousarray(proj, dtype=G1_PROJECTIVE).reshape(-1)
        out = np.zeros(proj.shape[0], dtype=G1_AFFINE)
        _lib.check(L.snarkvm_hip_g1_to_affine(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(proj.ctypes.data), ctypes.c_size_t(proj.shape[0])))
        return out

    checks = {}

    def check_results(res, scalars_used, label, shifts=None):
        """Result k of a (pipelined) run equals ITS closed form (sum_i s_k[i] (i + 1)) G, s_k = the scalar vector of step k
        (`scalars_used` rotated by shifts[k] positions); fails loudly otherwise."""
        got = to_affine(res)
        shifts = [0] * got.shape[0] if shifts is None else shifts
        wants = {}
        for k in range(got.shape[0]):
            if shifts[k] not in wants:
                sk = scalars_used if shifts[k] == 0 else np.roll(scalars_used, shifts[k], axis=0)
                wants[shifts[k]] = device_multiple_of_g(weighted_sum_mod_r(sk, start=1))
            if got[k : k + 1].tobytes() != wants[shifts[k]].tobytes():
                raise SystemExit(f"bench.py: RESULT MISMATCH in {label}, instance {k}: the timed MSM does not equal the closed form")
        checks[label] = f"{got.shape[0]} results == their own closed forms ({len(wants)} distinct scalar vectors; device one-point MSM)"
        return wants[shifts[0]]

    # ------------------------------------------------------------------ inputs (synthetic, resident in HBM)
    n = 1 << args.lg_msm
    bases_dev = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(bases_dev.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(n)))
    if args.table_bits < 0:
        # measured on MI355X (profiles/r01_size_sweep.md): 22-bit windows win from 2^24, 20-bit from 2^22, 16-bit below
        args.table_bits = 16 if args.tables else (22 if args.lg_msm >= 24 else 20 if args.lg_msm >= 22 else 16)
    if not args.tables:
        args.tables = 16 if args.table_bits == 16 else -(-254 // args.table_bits)
    t0 = time.perf_counter()
    rb = RegisteredBases(device_ptr=bases_dev.data_ptr(), npoints=n, tables=args.tables,
                         window_bits=0 if (args.table_bits == 16 and args.tables == 16) or args.tables == 1 else args.table_bits)
    registration_ms = (time.perf_counter() - t0) * 1e3
    scalars_uniform = synthetic.random_fr_integers(n, synthetic.SEED_MSM_LARGE + rank)
    scalars = synthetic.witness_like_from(scalars_uniform, synthetic.SEED_MSM_LARGE + rank) if args.scalars == "witness" else scalars_uniform
    d_scalars = torch.from_numpy(scalars.view(np.int64)).cuda()
    torch.cuda.synchronize()
    # Every timed step gets its OWN scalar vector (512 MiB each in HBM): step k uses the seeded vector rotated by k * STEP_SHIFT
    # positions, i.e. a different scalar for every base; each result is checked against its own closed form afterwards.
    STEP_SHIFT = 999983
    shifts = [(k * STEP_SHIFT) % n for k in range(args.steps)]
    d_step = [d_scalars if sh == 0 else torch.roll(d_scalars.view(n, 4), sh, dims=0).contiguous() for sh in shifts]
    torch.cuda.synchronize()

    # ------------------------------------------------------------------ MSM: W warm-up + K timed steps
    if args.no_pipeline:
        for _ in range(args.warmup):
            rb.msm(device_ptr=d_scalars.data_ptr(), npoints=n, window_bits=args.window_bits)
    elif args.warmup:
        # the pipelined path cycles through several HIP streams with one workspace each: W warm-up steps per stream, so that
        # no workspace is allocated (hipMalloc / hipFree synchronise the device) inside the timed region
        lanes = L.snarkvm_hip_batch_lanes(ctypes.c_size_t(n))
        rb.msm_batch(device_ptrs=[d_scalars.data_ptr()] * (lanes * args.warmup), npoints=[n] * (lanes * args.warmup), window_bits=args.window_bits)
    barrier()
    sclk = SclkSampler(dev_index)
    with sclk:
        t0 = time.perf_counter()
        if args.no_pipeline:
            res = np.concatenate([rb.msm(device_ptr=d.data_ptr(), npoints=n, window_bits=args.window_bits) for d in d_step])
        else:
            # K independent MSM instances (a batch of commitments) pipelined over the backend's HIP streams: the
            # latency-bound tail of one instance overlaps the accumulation of the next.  Every step does the full work.
            res = rb.msm_batch(device_ptrs=[d.data_ptr() for d in d_step], npoints=[n] * args.steps, window_bits=args.window_bits)
        dt_own = time.perf_counter() - t0  # this rank's own K steps, before it waits for the others
        barrier()
        dt_rank = time.perf_counter() - t0
        dt = max_over_ranks(dt_rank)
    rank_dts = gather_over_ranks(dt_own)
    print(f"[bench rank {rank}] {dt_rank / args.steps * 1e3:.3f} ms per step on this rank", file=sys.stderr)
    ms_per_step = dt / args.steps * 1e3
    pairs_per_s = world * n * args.steps / dt
    want_affine = check_results(res, scalars, "timed_msm", shifts)  # outside the timed region
    del d_step
    # ---- the same K pipelined steps on the OTHER distribution (rank 0, N = 1): SURVEY.md 8(d) config 2 names witness-like scalars as the second one
    other = None
    if rank == 0 and world == 1 and not args.no_extra_legs and not args.no_pipeline:
        other_name = "uniform" if args.scalars == "witness" else "witness_like"
        sc_o = scalars_uniform if args.scalars == "witness" else synthetic.witness_like_from(scalars_uniform, synthetic.SEED_MSM_LARGE + rank)
        d_o = torch.from_numpy(sc_o.view(np.int64)).cuda()
        k_o = min(args.steps, 8)
        sh_o = shifts[:k_o]
        d_step_o = [d_o if sh == 0 else torch.roll(d_o.view(n, 4), sh, dims=0).contiguous() for sh in sh_o]
        torch.cuda.synchronize()
        rb.msm_batch(device_ptrs=[d_o.data_ptr()] * L.snarkvm_hip_batch_lanes(ctypes.c_size_t(n)), npoints=[n] * L.snarkvm_hip_batch_lanes(ctypes.c_size_t(n)), window_bits=args.window_bits)
        barrier()
        t0 = time.perf_counter()
        res_o = rb.msm_batch(device_ptrs=[d.data_ptr() for d in d_step_o], npoints=[n] * k_o, window_bits=args.window_bits)
        barrier()
        dt_o = time.perf_counter() - t0
        check_results(res_o, sc_o, f"timed_msm_{other_name}", sh_o)
        other = {"name": other_name, "value": n * k_o / dt_o, "ms_per_step": dt_o / k_o * 1e3, "steps": k_o}
        del d_step_o, d_o, sc_o
    del scalars_uniform

    # ------------------------------------------------------------------ per-phase kernel times (HIP events on the launch stream)
    L.snarkvm_hip_set_profiling(1)
    phase_ms = {}
    reps = 3
    for _ in range(reps):
        rb.msm(device_ptr=d_scalars.data_ptr(), npoints=n, window_bits=args.window_bits)
        for i in range(L.snarkvm_hip_get_phase_count()):
            name = L.snarkvm_hip_get_phase_name(i).decode()
            phase_ms[name] = phase_ms.get(name, 0.0) + L.snarkvm_hip_get_phase_ms(i) / reps
    L.snarkvm_hip_set_profiling(0)

    # ------------------------------------------------------------------ NTT at 2^lg_ntt (device resident, in place)
    nn = 1 << args.lg_ntt
    x = synthetic.random_fr_integers(nn, synthetic.SEED_NTT + rank)  # any residues < r are valid Montgomery images
    d_x = torch.from_numpy(x.view(np.int64)).cuda()
    torch.cuda.synchronize()

    def ntt_dev(t, lg, direction):
        _lib.check(L.snarkvm_hip_ntt_device(ctypes.c_void_p(t.data_ptr()), ctypes.c_uint32(lg), 0, direction, 0))

    def time_ntt(t, lg, steps, batched=True):
        """`steps` transforms (forward / inverse alternating, so the vector ends as it began).  batched: ONE
        snarkvm_hip_ntt_device_batch call - one enqueue, one synchronisation, like the MSM's pipelined batch (a prover round
        issues its independent transforms this way); otherwise one synchronous snarkvm_hip_ntt_device call per transform."""
        ptrs = (ctypes.c_void_p * steps)(*([t.data_ptr()] * steps))
        dirs = (ctypes.c_int * steps)(*[i & 1 for i in range(steps)])
        # untimed warm-up: the same number of transforms (the host-side preparation before this leg leaves the GPU idle for
        # seconds; the first ~20 ms of work after an idle period run at a lower clock: 2.35 instead of 2.16 ms per 2^24 transform)
        _lib.check(L.snarkvm_hip_ntt_device_batch(ptrs, ctypes.c_size_t(steps), ctypes.c_uint32(lg), 0, dirs, None))
        barrier()
        t0 = time.perf_counter()
        if batched:
            _lib.check(L.snarkvm_hip_ntt_device_batch(ptrs, ctypes.c_size_t(steps), ctypes.c_uint32(lg), 0, dirs, None))
        else:
            for i in range(steps):
                ntt_dev(t, lg, i & 1)
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    if args.ntt_steps & 1:
        args.ntt_steps += 1  # forward / inverse pairs: the vector is back to the input afterwards
    # `ntt_value` = one synchronous snarkvm_hip_ntt_device call per transform: the reference's call pattern (EvaluationDomain::fft_in_place
    # -> snarkvm_ntt per vector), comparable with earlier rounds and with the CPU baseline; the same transforms as ONE batch call beside it
    ntt_batch_dt = time_ntt(d_x, args.lg_ntt, args.ntt_steps)
    ntt_dt = time_ntt(d_x, args.lg_ntt, args.ntt_steps, batched=False)
    ntt_elems_per_s = world * nn * args.ntt_steps / ntt_dt
    L.snarkvm_hip_set_profiling(1)
    ntt_dev(d_x, args.lg_ntt, 0)
    ntt_kernel_ms = L.snarkvm_hip_get_phase_ms(0)
    L.snarkvm_hip_set_profiling(0)
    ntt_forward = d_x.cpu().numpy().view(np.uint64).reshape(-1, 4)  # forward transform of x (for the oracle check below)
    ntt_dev(d_x, args.lg_ntt, 1)
    if not np.array_equal(d_x.cpu().numpy().view(np.uint64).reshape(-1, 4), x):
        raise SystemExit("bench.py: NTT round trip does not return the input")
    checks["ntt_round_trip"] = f"iNTT(NTT(x)) == x at 2^{args.lg_ntt}"

    # ------------------------------------------------------------------ extra legs (rank 0, N = 1): the rest of BASELINE.json's metric
    extra = {}
    if rank == 0 and world == 1 and not args.no_extra_legs:
        table_bytes = args.tables * n * 128
        extra["registration_ms"] = registration_ms
        extra["table_bytes"] = table_bytes
        # -- the same MSM without precomputed tables (registered bases only: 16 windows of 16 bits, Horner chain on the host)
        # (16 digit rows per scalar instead of 12: every lane the batch cycles through must GROW its workspace first.  Round 4 warmed one lane with a
        # synchronous call and timed a 2-instance batch, i.e. lane 1 outgrew six buffers - hipFree + hipMalloc, 2.8 GiB - inside the timed region, behind
        # lane 0's running MSM; the driver's round-4 run read 93.6 ms per step here, which round 5 could not reproduce, not even with round 4's library
        # (profiles/r05_summary.md).  The warm-up now is a batch of the same shape over every lane, the library no longer frees an outgrown buffer in the
        # middle of a call, and `tables1_workspace_growth` says whether anything grew inside the timed region: it must read zero allocations.)
        rb1 = RegisteredBases(device_ptr=bases_dev.data_ptr(), npoints=n, tables=1)
        k1 = 2
        lanes1 = max(k1, L.snarkvm_hip_batch_lanes(ctypes.c_size_t(n)))
        rb1.msm_batch(device_ptrs=[d_scalars.data_ptr()] * lanes1, npoints=[n] * lanes1)
        barrier()
        alloc_stats(L, reset=True)
        t0 = time.perf_counter()
        r1 = rb1.msm_batch(device_ptrs=[d_scalars.data_ptr()] * k1, npoints=[n] * k1)
        barrier()
        d1 = time.perf_counter() - t0
        extra["tables1_workspace_growth"] = alloc_stats(L)
        rb1.close()
        if to_affine(r1)[0:1].tobytes() != want_affine.tobytes():
            raise SystemExit("bench.py: RESULT MISMATCH in the tables = 1 leg")
        extra["tables1_value"] = n * k1 / d1
        extra["tables1_ms_per_step"] = d1 / k1 * 1e3
        # -- 2^20: MSM over 16 x 16-bit tables (pipelined batch of 8) and NTT
        if args.lg_msm >= 20:
            n20 = 1 << 20
            rb20 = RegisteredBases(device_ptr=bases_dev.data_ptr(), npoints=n20, tables=16)
            lanes = L.snarkvm_hip_batch_lanes(ctypes.c_size_t(n20))
            rb20.msm_batch(device_ptrs=[d_scalars.data_ptr()] * lanes, npoints=[n20] * lanes)
            barrier()
            t0 = time.perf_counter()
            k20 = 16
            slices = [(k % (n // n20)) * n20 for k in range(k20)]  # 16 different scalar vectors: consecutive 2^20-slices of the seeded vector
            r20 = rb20.msm_batch(device_ptrs=[d_scalars.data_ptr() + 32 * o for o in slices], npoints=[n20] * k20)
            barrier()
            d20 = time.perf_counter() - t0
            a20 = to_affine(r20)
            for k, o in enumerate(slices):
                if a20[k : k + 1].tobytes() != device_multiple_of_g(weighted_sum_mod_r(scalars[o : o + n20], start=1)).tobytes():
                    raise SystemExit(f"bench.py: RESULT MISMATCH in msm_2p20, instance {k}")
            checks["msm_2p20"] = f"{k20} results == their own closed forms ({len(set(slices))} distinct scalar vectors)"
            t0 = time.perf_counter()
            for _ in range(5):
                rb20.msm(device_ptr=d_scalars.data_ptr(), npoints=n20)
            s20 = (time.perf_counter() - t0) / 5
            rb20.close()
            extra["msm_2p20"] = {"value": n20 * k20 / d20, "unit": "pairs/s", "ms_per_step_pipelined": d20 / k20 * 1e3, "ms_sync": s20 * 1e3, "base_tables": "16 x 16 bit"}
        if args.lg_ntt >= 20:
            nt = 20
            d20b = time_ntt(d_x, 20, nt)
            d20 = time_ntt(d_x, 20, nt, batched=False)
            extra["ntt_2p20"] = {"value": (1 << 20) * nt / d20, "unit": "elements/s", "ms_per_transform": d20 / nt * 1e3, "ms_per_transform_one_batch_call": d20b / nt * 1e3}
        # -- the reference's own FFI symbols over host buffers (PCIe-inclusive; never `value`)
        ffi = {}
        host_bases = bases_dev.cpu().numpy().view(G1_AFFINE)
        for lg in (16, 20, args.lg_msm):
            m = 1 << lg
            hb, hs = host_bases[:m], scalars[:m]
            warm = hb.copy()
            plugin.msm(warm, hs)          # another host range of the same size: workspaces and pinned staging exist from here on
            fresh = hb.copy()
            del warm
            t0 = time.perf_counter()
            plugin.msm(fresh, hs)         # sample 1: a host buffer the HIP runtime never copied from
            firsts = [time.perf_counter() - t0]
            del fresh
            t0 = time.perf_counter()
            r_first = plugin.msm(hb, hs)  # sample 2: a buffer the runtime filled itself (D2H)
            firsts.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            r_again = plugin.msm(hb, hs)  # the symbol is stateless: a repeated call costs the same (upload + conversion + table-less MSM)
            firsts.append(time.perf_counter() - t0)
            first = min(firsts)
            # the EXTENSION a caller opts into (rust/: resident::Bases): the SRS registered once with precomputed tables, host scalars per call
            tb, bits = (12, 22) if lg >= 24 else (13, 20) if lg >= 21 else (16, 0) if lg >= 18 else (17, 15)
            rbx = rb if (lg == args.lg_msm and tb == args.tables) else RegisteredBases(device_ptr=bases_dev.data_ptr(), npoints=m, tables=tb, window_bits=bits)
            rbx.msm(hs)
            t0 = time.perf_counter()
            r_reg = rbx.msm(hs)
            reg = time.perf_counter() - t0
            if rbx is not rb:
                rbx.close()
            w = device_multiple_of_g(weighted_sum_mod_r(hs, start=1))
            if any(to_affine(r).tobytes() != w.tobytes() for r in (r_first, r_again, r_reg)):
                raise SystemExit(f"bench.py: RESULT MISMATCH in snarkvm_msm at 2^{lg}")
            ffi[f"snarkvm_msm_2p{lg}"] = {"call_ms": first * 1e3, "call_ms_samples": [f * 1e3 for f in firsts], "pairs_per_s": m / first,
                                         "registered_bases_host_scalars_ms": reg * 1e3, "registered_pairs_per_s": m / reg, "registered_tables": f"{tb} x {bits or 16} bit"}
        from snarkvm_amd.layout import NTTDirection, NTTInputOutputOrder, NTTType
        for lg in (16, 20, args.lg_ntt):
            y = x[: 1 << lg].copy()
            plugin.NTT(1 << lg, y, NTTInputOutputOrder.NN, NTTDirection.Forward, NTTType.Standard)
            t0 = time.perf_counter()
            plugin.NTT(1 << lg, y, NTTInputOutputOrder.NN, NTTDirection.Inverse, NTTType.Standard)
            d = time.perf_counter() - t0
            if not np.array_equal(y, x[: 1 << lg]):
                raise SystemExit(f"bench.py: snarkvm_ntt round trip failed at 2^{lg}")
            ffi[f"snarkvm_ntt_2p{lg}"] = {"ms": d * 1e3, "elements_per_s": (1 << lg) / d}
        for lg in (16, 20, args.lg_ntt):  # PolyMultiplier::multiply of two coefficient vectors of 2^(lg-1) elements on the 2^lg domain
            m = 1 << lg
            ops = [np.ascontiguousarray(x[: m // 2]), np.ascontiguousarray(x[m // 2 : m])]
            prod = np.zeros((m, 4), dtype=np.uint64)
            prod[:] = 0  # touched pages, like Rust's vec![zero; domain] (lib.rs:126-127)
            pp = (ctypes.c_void_p * 2)(*[o.ctypes.data for o in ops])
            pl = (ctypes.c_size_t * 2)(*[o.shape[0] for o in ops])
            args_c = (ctypes.c_void_p(prod.ctypes.data), ctypes.c_size_t(2), pp, pl, ctypes.c_size_t(0), None, None, ctypes.c_uint32(lg))
            _lib.check(L.snarkvm_polymul(*args_c))
            t0 = time.perf_counter()
            _lib.check(L.snarkvm_polymul(*args_c))
            d = time.perf_counter() - t0
            ffi[f"snarkvm_polymul_2p{lg}"] = {"ms": d * 1e3, "operands": 2, "pcie_bytes": 64 * m}
        ffi["note"] = ("host buffers in and out through the reference's FFI symbols.  snarkvm_msm is stateless (nothing retained between calls): call_ms = "
                       "upload of bases + scalars (2.4 GB over PCIe at 2^24) overlapped with conversion and a table-less MSM in 2^20-pair chunks (first chunk ramped, last tapered, one bucket sink).  "
                       "registered_bases_host_scalars_ms = the extension ABI (snarkvm_hip_register_bases_windowed once, snarkvm_hip_msm_registered per call: "
                       "only the scalars cross PCIe)")
        extra["end_to_end_ffi"] = ffi
        checks["ffi"] = "snarkvm_msm (stateless) and registered-bases calls == closed form at every size; snarkvm_ntt round trips"
        del host_bases
    del bases_dev
    if rank == 0 and world == 1 and not args.no_extra_legs and not args.no_proof_legs:
        torch.cuda.empty_cache()
        extra.update(proof_legs(dev_index, with_oracle=not args.no_cpu_baseline))

    # ------------------------------------------------------------------ CPU baseline + oracle checks (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu as oracle

        oracle_build = oracle.use_native()
        cores = os.cpu_count() or 1
        threads = min(cores, 64)
        oracle.set_threads(threads)
        cn = min(1 << args.cpu_lg_msm, n)
        ogen = np.zeros(1, dtype=oracle.G1_AFFINE)
        ogen["x"] = G1_GEN_X
        ogen["y"] = G1_GEN_Y
        cb = oracle.g1_gen_bases(ogen, 1, cn)
        cs = scalars[:cn]
        oracle.g1_msm(cb[:1024], cs[:1024])  # warm-up
        t0 = time.perf_counter()
        cpu_res = oracle.g1_msm(cb, cs, oracle.MSM_BATCHED)
        cpu_msm_dt = time.perf_counter() - t0
        # the oracle pins the closed form the timed results were checked against, and the CPU sample itself
        k_full = weighted_sum_mod_r(scalars, start=1)
        o_full = oracle.g1_to_affine(oracle.g1_mul(ogen, np.array([(k_full >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)))
        if o_full.tobytes() != want_affine.tobytes():
            raise SystemExit("bench.py: the timed MSM result differs from the CPU oracle")
        k_cn = weighted_sum_mod_r(cs, start=1)
        o_cn = oracle.g1_to_affine(oracle.g1_mul(ogen, np.array([(k_cn >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)))
        if oracle.g1_to_affine(cpu_res).tobytes() != o_cn.tobytes():
            raise SystemExit("bench.py: the CPU oracle disagrees with itself (batched::msm vs closed form)")
        checks["timed_msm_vs_oracle"] = f"2^{args.lg_msm} result == oracle (batched::msm restatement's scalar multiplication of the closed form)"
        cnn = min(1 << args.cpu_lg_ntt, nn)
        t0 = time.perf_counter()
        o_ntt = oracle.ntt(x[:cnn].copy())
        cpu_ntt_dt = time.perf_counter() - t0
        if cnn == nn:
            if not np.array_equal(o_ntt, ntt_forward):
                raise SystemExit("bench.py: the device NTT output differs from the CPU oracle")
            checks["ntt_vs_oracle"] = f"forward NTT 2^{args.lg_ntt} bit-exact vs oracle.ntt (all {nn} elements)"
        cpu = {
            "value": cn / cpu_msm_dt,
            "unit": "pairs/s",
            "cores": threads,
            "kind": "port",
            "sample": f"batched::msm restatement (oracle/cpu_oracle.cpp, OpenMP one task per window) on {cn} pairs of the "
                      f"same workload: {cpu_msm_dt:.2f} s; fft_in_place restatement on {cnn} elements: {cpu_ntt_dt:.2f} s",
            "host_cores": cores,
            "build": f"g++ {oracle_build}, OpenMP",
            "ntt_value": cnn / cpu_ntt_dt,
            "ntt_unit": "elements/s",
        }

    # HBM bytes / instruction counts per dispatch from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE
    # runs; gfx950 x2 FETCH correction), and the wall-clock arithmetic ceilings of tools/ecbench.hip / tools/microbench.hip.
    # Only quoted for the configuration they were collected on; PMC collection cannot run inside this process.
    default_cfg = args.lg_msm == 24 and args.lg_ntt == 24 and args.tables == 12 and args.table_bits == 22 and not args.window_bits
    pmc, pmc_source = {}, None
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        if default_cfg and load_profile_json(name):
            pmc = load_profile_json(name).get("kernels", {})
            pmc_source = f"look-up, not measured in this run: profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration on the builder's box)"
            break
    ceil = load_profile_json("r03_alu_ceilings.json") or load_profile_json("r02_alu_ceilings.json")

    def traffic(kernel_prefix, fetch_key, times=1):
        for name, k in pmc.items():
            if name.startswith(kernel_prefix):
                return (k[fetch_key] + k["write_bytes"]) * times
        return None

    if rank == 0:
        acc_ms = phase_ms.get("msm_accumulate", 0.0)
        dig_ms = phase_ms.get("msm_scalar_read", 0.0) or phase_ms.get("msm_digits", 0.0)
        dig_kernel = "radix_hist1_wide_kernel (scalar read + level-1 histograms; no digit matrix)" if "msm_scalar_read" in phase_ms else "msm_digits_kernel"
        cbits = args.window_bits or (args.table_bits if args.tables > 1 else 16)
        W = args.tables * (args.table_bits // cbits) if args.tables > 1 else (254 + cbits - 1) // cbits  # digit rows per scalar
        alg_bytes = n * 128.0 + 144.0  # whole MSM = n (32 + 96) + 144
        madds = float(n) * W           # mixed additions of the accumulate kernel (one per digit; digit 0 is rare)
        alu = None
        if ceil and acc_ms:
            alu = {
                "kernel": "msm_accumulate_lazy_kernel",
                "madds_per_launch": madds,
                "madds_per_s": madds / (acc_ms * 1e-3),
                "madd_ceiling_per_s": ceil.get("g1_lazy_madd_per_s") or ceil.get("g1_madd_per_s"),
                # (round 6: was `frac`.  The ceiling is the kernel's OWN addition loop run from registers - a self-reference, not a roofline; `mad_frac` below, against
                # the chip's measured v_mad_u64_u32 issue rate, is the ALU roofline fraction)
                "vs_register_loop": madds / (acc_ms * 1e-3) / (ceil.get("g1_lazy_madd_per_s") or ceil["g1_madd_per_s"]) if ceil.get("g1_madd_per_s") else None,
                "mads_per_launch": madds * float(ceil.get("g1_lazy_mads_per_madd", 2938)),
                "peak_mads_per_s": ceil.get("v_mad_u64_u32_per_s"),
                "mad_frac": madds * float(ceil.get("g1_lazy_mads_per_madd", 2938)) / (acc_ms * 1e-3) / ceil["v_mad_u64_u32_per_s"] if ceil.get("v_mad_u64_u32_per_s") else None,
                "source": "look-up, not measured in this run: profiles/r03_alu_ceilings.json (tools/ecbench.hip k_lazy at the kernel's occupancy, tools/microbench.hip: wall-clock, whole chip, on the builder's box)",
                # the shader clock sampled by a host thread during the timed MSM steps (amdgpu hwmon): the ceilings above assume the 2.4 GHz
                # peak clock, so mad_frac against what the chip could issue at the clock it actually sustained is mad_frac * 2400 / mean_mhz
                "sclk_during_timed_steps": sclk.summary(),
            }
            if alu["sclk_during_timed_steps"] and alu.get("mad_frac"):
                alu["mad_frac_at_sustained_clock"] = alu["mad_frac"] * 2400.0 / alu["sclk_during_timed_steps"]["mean_mhz"]
        alu_ntt = None
        if ceil and ntt_kernel_ms and ceil.get("v_mad_u64_u32_per_s"):
            # Fr products per element: lg / 2 butterfly products + one closing product per non-last pass + the final reduction (~half)
            prods = nn * (args.lg_ntt / 2.0 + 2.0 + 0.5)
            alu_ntt = {"fr_products_per_launch": prods, "mads_per_launch": prods * 153.0, "peak_mads_per_s": ceil["v_mad_u64_u32_per_s"],
                       "mad_frac": prods * 153.0 / (ntt_kernel_ms * 1e-3) / ceil["v_mad_u64_u32_per_s"]}
            if ceil.get("fr_ntt_element_passes_per_s"):
                # the same accounting as the accumulate kernel: work / time / the register-resident rate of the kernel's own arithmetic
                passes = (args.lg_ntt + 7) // 8 if args.lg_ntt <= 24 else 3  # ntt_make_plan: radix <= 2^8 up to 2^24, 2^9 beyond
                alu_ntt["element_passes_per_launch"] = float(passes * nn)
                alu_ntt["element_pass_ceiling_per_s"] = ceil["fr_ntt_element_passes_per_s"]
                alu_ntt["vs_register_loop"] = passes * nn / (ntt_kernel_ms * 1e-3) / ceil["fr_ntt_element_passes_per_s"]
        out = {
            "metric": "BLS12-377 G1 MSM scalar-point pairs/sec (+ Fr NTT elements/sec in ntt_*)",
            "value": pairs_per_s,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "rank_ms_per_step": [d / args.steps * 1e3 for d in rank_dts],
            "rank_devices": rank_devices,
            # DESIGN.md 7's prediction made checkable: the sum of every rank's OWN rate (its K steps on its own clock, before the closing barrier) - what N
            # independent instance streams deliver when nothing on the host or the fabric couples them.  `value` (all ranks' units / the slowest rank's time incl. the
            # barrier) should sit within a few percent of it; a gap means host-side contention, two ranks on one GPU (duplicate_devices) or a slow box.
            "predicted_value": sum(n * args.steps / d for d in rank_dts),
            "value_over_predicted": pairs_per_s / sum(n * args.steps / d for d in rank_dts),
            "duplicate_devices": duplicate_devices(rank_devices),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,  # the reference publishes no number for this metric
            "vs_cpu_baseline": (pairs_per_s / cpu["value"]) if cpu else None,
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": f"G1 Pippenger MSM 2^{args.lg_msm} (BASELINE.json configs[1]), bases (i+1)G registered in HBM, "
                                   f"{'witness-like (50 % zero, 25 % < 2^16, 25 % uniform)' if args.scalars == 'witness' else 'uniform'} scalars in HBM; independent instance per GPU",
                       "scalars": args.scalars,
                       "lg_msm": args.lg_msm, "lg_ntt": args.lg_ntt, "window_bits": args.window_bits or "auto", "base_tables": args.tables, "table_bits": args.table_bits,
                       "pipelined_batch": not args.no_pipeline},
            "checks": checks,
            # the same pipelined steps on the second distribution (rank 0, N = 1), each result against its own closed form
            **({f"value_{other['name']}": other["value"], f"ms_per_step_{other['name']}": other["ms_per_step"], f"steps_{other['name']}": other["steps"],
                f"{other['name']}_over_headline": other["value"] / pairs_per_s} if other else {}),
            "ntt_value": ntt_elems_per_s,
            "ntt_unit": "elements/s",
            "ntt_ms_per_transform": ntt_dt / args.ntt_steps * 1e3,
            "ntt_batched_call": {"value": world * nn * args.ntt_steps / ntt_batch_dt, "unit": "elements/s", "ms_per_transform": ntt_batch_dt / args.ntt_steps * 1e3,
                                 "what": "the same transforms as ONE snarkvm_hip_ntt_device_batch call (one enqueue, one synchronisation); ntt_value: one synchronous "
                                         "snarkvm_hip_ntt_device call per transform"},
            "ntt_kernel_ms": ntt_kernel_ms,
            "ntt_vs_cpu_baseline": (ntt_elems_per_s / cpu["ntt_value"]) if cpu else None,
            "phase_ms": {k: round(v, 4) for k, v in phase_ms.items()},
            # dominant kernel: bucket accumulation.  SURVEY.md 8(d): algorithmic bytes of the whole MSM = n (32 + 96) + 144; the
            # kernel is integer-ALU bound, so `frac` is small by nature - `alu_roofline` is the bound that applies.
            "roofline": {
                "bound": "hbm",
                "kernel": "msm_accumulate_lazy_kernel",
                "achieved": alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms else None,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": (alg_bytes / (acc_ms * 1e-3) / 1e9 / 8000.0) if acc_ms else None,
                "traffic": traffic("msm_accumulate_lazy_kernel", "fetch_bytes_raw") or traffic("msm_accumulate_seg_kernel", "fetch_bytes_raw"),
                "traffic_source": pmc_source,
                "algorithmic_bytes": alg_bytes,
                "traffic_model": {"bytes": n * 132.0 * W, "what": "one gathered 128-B lazy base slot + one 4-B sorted index per (pair, digit row)"},
                "note": "whole-MSM is integer-ALU bound (SURVEY.md 8d): see alu_roofline; roofline_scalar_read is the HBM-bound phase",
            },
            "alu_roofline": alu,
            # the phase north_star scopes the HBM claim to: scalar read + digit extraction, 32 B per scalar
            "roofline_scalar_read": {
                "bound": "hbm",
                "kernel": dig_kernel,
                "achieved": (32.0 * n) / (dig_ms * 1e-3) / 1e9 if dig_ms else None,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": ((32.0 * n) / (dig_ms * 1e-3) / 1e9 / 8000.0) if dig_ms else None,
                "traffic": (traffic("radix_hist1_wide_kernel", "fetch_bytes_x2") or traffic("radix_hist1_fused_kernel", "fetch_bytes_x2")) if "msm_scalar_read" in phase_ms else traffic("msm_digits_kernel", "fetch_bytes_x2"),
                "traffic_source": pmc_source,
                "algorithmic_bytes": 32.0 * n,
                # the whole scalar-consuming phase: the level-1 scatter (radix_scatter1_fused_kernel + its counter scans) reads every
                # scalar a second time and writes 72 B of (index | sign, remainder) entries per scalar; still priced on 32 n bytes
                "whole_phase": ({
                    "kernels": "radix_hist1_wide + fused_chunk_sums / scan / fused_tile_offsets + radix_scatter1_fused",
                    "ms": dig_ms + phase_ms["msm_sort_level1"],
                    "achieved": (32.0 * n) / ((dig_ms + phase_ms["msm_sort_level1"]) * 1e-3) / 1e9,
                    "frac": (32.0 * n) / ((dig_ms + phase_ms["msm_sort_level1"]) * 1e-3) / 1e9 / 8000.0,
                    "bytes_moved_model": (32.0 + 32.0 + 72.0 + 3.0) * n,
                    "frac_on_bytes_moved": (139.0 * n) / ((dig_ms + phase_ms["msm_sort_level1"]) * 1e-3) / 1e9 / 8000.0,
                } if dig_ms and "msm_scalar_read" in phase_ms and phase_ms.get("msm_sort_level1") else None),
            },
            "roofline_ntt": {
                "bound": "hbm",
                "kernel": "ntt_pass_kernel_v2 (3 passes)",
                "achieved": (64.0 * nn) / (ntt_kernel_ms * 1e-3) / 1e9 if ntt_kernel_ms else None,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": ((64.0 * nn) / (ntt_kernel_ms * 1e-3) / 1e9 / 8000.0) if ntt_kernel_ms else None,
                "traffic": traffic("ntt_pass_kernel_v2", "fetch_bytes_x2", times=3),
                "traffic_source": pmc_source,
                "algorithmic_bytes": 64.0 * nn,
                "alu_roofline": alu_ntt,
            },
            "cpu_baseline": cpu,
        }
        out.update(extra)
        print(json.dumps(out))
    rb.close()
    if world > 1:
        dist.destroy_process_group()


def oracle_check_proof(keys, shape, got, p):
    """All 15 results of proof `p` (`got`: 14 G1Projective records + the G2Projective record, as bytes) against oracle/proof_replay.py;
    raises SystemExit on a mismatch, returns the oracle's seconds.  Test infrastructure: only ever called outside timed regions."""
    from oracle import cpu as oracle
    from oracle import proof_replay
    from snarkvm_amd import kzg10
    from snarkvm_amd.layout import G1_PROJECTIVE, G2_PROJECTIVE

    oracle.use_native()  # (before the first oracle call of the process: the cpu_baseline leg times this library)
    oracle.set_threads(min(os.cpu_count() or 1, 64))
    t0 = time.perf_counter()
    want = proof_replay.expected_results(keys.pool_host, keys.g1_host, keys.g2_host, keys.point, shape.lg_r, shape.lg_k, shape.lg_g2, shape.nmax, p)
    oracle_s = time.perf_counter() - t0
    for j in range(14):
        ga = kzg10.to_affine(np.frombuffer(got[j], dtype=G1_PROJECTIVE))
        wa = want[j]
        if not (np.array_equal(ga["x"], wa["x"]) and np.array_equal(ga["y"], wa["y"]) and np.array_equal(ga["infinity"], wa["infinity"])):
            raise SystemExit(f"bench.py: proof {p}, commitment {j}: the device result differs from the CPU oracle")
    if want[14] is not None:
        g2a = oracle.g2_to_affine(np.frombuffer(got[14], dtype=G2_PROJECTIVE))
        if g2a.tobytes() != want[14].tobytes():
            raise SystemExit(f"bench.py: proof {p}: the G2 MSM result differs from the CPU oracle")
    return oracle_s


def alloc_stats(L, reset=False):
    v = (ctypes.c_uint64 * 5)()
    L.snarkvm_hip_alloc_stats(v, 1 if reset else 0)
    return {"device_allocations": int(v[0]), "device_bytes": int(v[1]), "pinned_allocations": int(v[2]), "pinned_bytes": int(v[3]), "ms": v[4] / 1e3}


def proof1_run(keys, dev_index, salts, warm=4, async_msm=True, await_rounds=False, msm_in_stream=False):
    """`salts`: the proofs, proved ONE AT A TIME by this thread (snarkvm_amd/proofs.py::replay_single).  Returns (seconds of the whole
    run, per-proof latencies, per-proof result lists, call-time split, workspace growth inside the timed region)."""
    from snarkvm_amd import _lib, proofs

    L = _lib.lib()
    ws = proofs.SingleProofWorkspace(keys, dev_index)
    for sidx in salts[:warm]:
        proofs.replay_single(ws, sidx, None, async_msm, None, await_rounds, msm_in_stream)
    ws.times = {k: 0.0 for k in ws.times}
    _lib.check(L.snarkvm_hip_synchronize())
    alloc_stats(L, reset=True)
    lat, results = [], []
    t_begin = time.perf_counter()
    for sidx in salts:
        got = []
        t0 = time.perf_counter()
        proofs.replay_single(ws, sidx, got, async_msm, None, await_rounds, msm_in_stream)
        lat.append(time.perf_counter() - t0)
        results.append(got)
    dt = time.perf_counter() - t_begin
    return dt, lat, results, dict(ws.times), alloc_stats(L)


def proof1_summary(dt, lat, times, grown, count):
    ls = sorted(lat)
    return {"proofs": count, "ms_per_proof": dt / count * 1e3, "median_ms": ls[len(ls) // 2] * 1e3, "min_ms": ls[0] * 1e3, "max_ms": ls[-1] * 1e3,
            "host_enqueue_ms_per_proof": times["enqueue"] / count * 1e3, "host_wait_ms_per_proof": times["wait"] / count * 1e3,
            "workspace_growth_in_timed_region": grown}


def proof1_ffi_rows(keys, salts, threads, warm=3):
    """The transfer_private call list through the three symbols `snarkvm-algorithms-cuda` binds, on host buffers (snarkvm_amd/proofs.py::replay_ffi), one
    proof at a time: row `stateless` (the symbols as the reference defines them: nothing retained between calls) and row `base_cache_16`
    (snarkvm_hip_set_base_cache(16) == SNARKVM_HIP_BASE_CACHE=16: the long-lived base vector is registered in HBM at its second sighting).
    Returns ({row: summary}, {row: [per-proof result lists]}).  Timed per step class inside replay_ffi; the rows' results are checked by the caller."""
    from snarkvm_amd import _lib, proofs

    L = _lib.lib()
    host = proofs.FfiProofHost(keys, threads)
    rows, results = {}, {}
    for name, tables in (("stateless", 0), ("base_cache_16", 16)):
        _lib.check(L.snarkvm_hip_set_base_cache(tables))
        for p in salts[:warm]:  # the cache registers a range at its second sighting: from the third proof on every slice is a hit
            proofs.replay_ffi(host, p, None, {})
        per, walls, got_all = [], [], []
        for p in salts:
            t, got = {}, []
            t0 = time.perf_counter()
            proofs.replay_ffi(host, p, got, t)
            walls.append(time.perf_counter() - t0)
            per.append(t)
            got_all.append(got)
        n = len(salts)
        mean = lambda k: sum(x[k] for x in per) / n * 1e3  # noqa: E731
        inside = sorted((x["ntt"] + x["polymul"] + x["msm"]) * 1e3 for x in per)
        rows[name] = {"ms_inside_the_three_symbols_per_proof": sum(inside) / n, "median_ms": inside[n // 2], "min_ms": inside[0], "max_ms": inside[-1],
                      "ms_by_step": {"transforms_and_per_matrix_steps": mean("ntt"), "rowcheck_product": mean("polymul"), "commitment_rounds": mean("msm")},
                      "calls_per_proof": {"snarkvm_ntt": per[0]["ntt_calls"], "snarkvm_polymul": per[0]["polymul_calls"], "snarkvm_msm": per[0]["msm_calls"]},
                      "g2_extension_call_ms": mean("g2"), "glue_ms_not_counted": mean("glue"), "wall_ms_per_proof_with_glue": sum(walls) / n * 1e3,
                      "proofs": n, "caller_threads": threads}
        results[name] = got_all
    _lib.check(L.snarkvm_hip_set_base_cache(0))
    host.close()
    return rows, results


FFI_WHAT = ("the transfer_private call list through snarkvm_ntt / snarkvm_polymul / snarkvm_msm on HOST buffers (what an unmodified snarkVM build gets): every commitment one "
            "snarkvm_msm over a slice of ONE long-lived base vector, a round's commitments from caller threads (sonic_pc/mod.rs:186-245); the reference's CPU glue between the "
            "calls (convert_to_bigints, divisions, <= 3-point hiding MSMs) is not counted")


def proof1_ffi(args, dev_index):
    """`--workload proof1 --ffi-only`: BASELINE.json configs[3] as an UNMODIFIED snarkVM would run it - the three reference symbols on host slices - beside the
    resident path (device-resident operands, registered SRS, one scope per proof, commitments awaited in transcript order)."""
    from snarkvm_amd import proofs

    shape = proofs.ProofShape()
    ptab, pbits = (int(v) for v in args.proof_geometry.split("x"))
    keys = proofs.ProverKeys(shape, tables=ptab, window_bits=pbits, scalars=args.scalars)
    count = 16
    salts = list(range(count))
    t0 = time.perf_counter()
    rows, res = proof1_ffi_rows(keys, salts, args.ffi_threads)
    dt_rows = time.perf_counter() - t0
    dt_res, lat_res, got_res, times_res, _ = proof1_run(keys, dev_index, salts, async_msm=True, await_rounds=True, msm_in_stream=True)
    # the like-for-like row: the reference prover issues no G2 MSM at all, and the three-symbol figures above do not count one
    keys0 = proofs.ProverKeys(proofs.ProofShape(lg_g2=0), tables=ptab, window_bits=pbits)
    dt_res0, lat_res0, _, times_res0, _ = proof1_run(keys0, dev_index, salts, async_msm=True, await_rounds=True, msm_in_stream=True)
    keys0.close()
    # ---- checks (outside the timed regions)
    checks = {}
    norm_res = [proofs.normalize_results(r) for r in got_res]
    for name, got in res.items():
        if [proofs.normalize_results(r) for r in got] != norm_res:
            raise SystemExit(f"bench.py: --ffi-only row {name}: a result differs from the resident replay")
    checks["every_row_vs_resident_replay"] = f"all {count} proofs x 15 results of both FFI rows == the resident one-scope replay (after affine normalisation)"
    if not args.no_cpu_baseline:
        secs = [oracle_check_proof(keys, shape, res[name][-1], salts[-1]) for name in res]
        checks["proof_vs_oracle"] = f"proof {salts[-1]} of each FFI row: all 14 G1 commitments / openings and the G2 MSM == oracle/proof_replay.py ({secs[0]:.1f} s on the host each)"
    resident = proof1_summary(dt_res, lat_res, times_res, None, count)
    resident0 = proof1_summary(dt_res0, lat_res0, times_res0, None, count)
    st, ca = rows["stateless"], rows["base_cache_16"]
    print(json.dumps({
        "metric": "milliseconds one Varuna-proof-shaped call list spends inside the reference's three FFI symbols (BASELINE.json configs[3], unmodified snarkVM)",
        "value": st["ms_inside_the_three_symbols_per_proof"], "unit": "ms/proof", "n_gpus": 1, "steps": count, "warmup": 3,
        "ms_per_step": st["ms_inside_the_three_symbols_per_proof"], "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": FFI_WHAT, "caller_threads": args.ffi_threads, "proofs": count},
        "stateless": st, "base_cache_16": ca,
        "resident": dict(resident, what="device-resident operands + registered SRS + one scope per proof, commitments awaited in transcript order in-stream, G2 MSM included "
                                          f"(registered_srs {ptab} x {pbits})"),
        "resident_without_g2": dict(resident0, what="the same without the G2 MSM: the like-for-like row (the reference prover issues none, and the FFI rows do not count theirs)"),
        "ratios": {"stateless_over_resident_without_g2": st["ms_inside_the_three_symbols_per_proof"] / resident0["ms_per_proof"],
                   "base_cache_over_resident_without_g2": ca["ms_inside_the_three_symbols_per_proof"] / resident0["ms_per_proof"],
                   "stateless_over_resident": st["ms_inside_the_three_symbols_per_proof"] / resident["ms_per_proof"],
                   "base_cache_over_resident": ca["ms_inside_the_three_symbols_per_proof"] / resident["ms_per_proof"]},
        "wall_s_of_both_rows": dt_rows, "checks": checks}))
    keys.close()


def proof1(args, rank, world, dev_index, barrier, max_over_ranks, gather_over_ranks, rank_devices):
    """BASELINE.json configs[3]: one Varuna-proof-shaped call list at a time from ONE caller thread, 32 timed proofs per rank (the
    reference proves one transaction at a time: synthesizer/snark/src/proving_key/mod.rs:37 -> varuna.rs:336).  `ms_per_step` is the
    latency of one proof IN THE ORDER A PROVER IS BOUND TO: the commitments of round k are in the host's hands (snarkvm_hip_scope_collect) before
    round k + 1 is issued - they enter the Fiat-Shamir transcript that yields the next challenge (varuna.rs:336 ff.) - with the awaited rounds on the scope's
    own stream and the independent G2 MSM on a further one.  The replay that enqueues all six rounds at once (round 5's headline; an order no prover can
    use) is reported beside it as the library's ceiling.  Every timed result is compared with the serial replay (one synchronous call per step:
    `proofs.replay`), two whole proofs with the CPU oracle."""
    import torch.distributed as dist

    from snarkvm_amd import proofs

    if args.ffi_only:
        if world > 1:
            raise SystemExit("bench.py: --ffi-only is a single-GPU leg")
        return proof1_ffi(args, dev_index)
    shape = proofs.ProofShape()
    ptab, pbits = (int(v) for v in args.proof_geometry.split("x"))
    keys = proofs.ProverKeys(shape, tables=ptab, window_bits=pbits, scalars=args.scalars, mem=args.proof_mem)
    count = 32
    mine = list(range(rank, count * world, world))
    checks = {}
    modes = {"awaited_in_stream": dict(async_msm=True, await_rounds=True, msm_in_stream=True),
             "awaited_on_further_streams": dict(async_msm=True, await_rounds=True),
             "all_rounds_enqueued_at_once": dict(async_msm=True),
             "synchronous_commitments": dict(async_msm=False)}
    head = "synchronous_commitments" if args.proof1_sync_msm else ("all_rounds_enqueued_at_once" if args.proof1_all_at_once else "awaited_in_stream")
    barrier()
    t0 = time.perf_counter()
    dt_rank, lat, got, times, grown = proof1_run(keys, dev_index, mine, **modes[head])
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    # proof1_run's warm-up sits inside [t0, now): the per-proof figures come from its own clock, the whole-job rate from the slowest rank's
    dt_job = max_over_ranks(dt_rank)
    rank_dts = gather_over_ranks(dt_rank)
    others = {}
    for name, kw in modes.items():
        if name != head:
            _, lat_o, got_o, times_o, _ = proof1_run(keys, dev_index, mine, **kw)
            others[name] = (lat_o, got_o, times_o)
    # ---- checks (outside the timed regions)
    ref_ws = proofs.ProofWorkspace(keys, dev_index)
    for i, p in enumerate(mine):
        ref = []
        proofs.replay(ref_ws, p, ref)
        nref = proofs.normalize_results(ref)
        if nref != proofs.normalize_results(got[i]) or any(nref != proofs.normalize_results(o[1][i]) for o in others.values()):
            raise SystemExit(f"bench.py: proof {p}: the one-scope replay differs from the serial replay")
    checks["every_proof_vs_serial_replay"] = (f"all {len(mine)} timed proofs x 15 results (commitments awaited round by round in-stream and on further streams, all rounds enqueued at once, "
                                              f"synchronous commitments) == one synchronous call per step")
    if rank == 0 and not args.no_cpu_baseline:
        secs = [oracle_check_proof(keys, shape, got[mine.index(p)], p) for p in (mine[0], mine[-1])]
        checks["proofs_vs_oracle"] = (f"proofs {mine[0]} and {mine[-1]}: all 14 G1 commitments / openings and the G2 MSM == oracle/proof_replay.py "
                                      f"({secs[0]:.1f} + {secs[1]:.1f} s on the host)")
    if rank == 0:
        what = {"awaited_in_stream": "one SNARKVM_HIP_SCOPE_ASYNC_MSM scope per proof; the G2 MSM on a further stream, then snarkvm_hip_scope_set_flags(... | SNARKVM_HIP_SCOPE_MSM_IN_STREAM): "
                                     "every commitment round on the scope's own stream and COLLECTED (snarkvm_hip_scope_collect) before the next round is issued - the Fiat-Shamir order",
                "awaited_on_further_streams": "the same order, the awaited rounds on further streams of the scope (an event hand-off per round)",
                "all_rounds_enqueued_at_once": "all six commitment rounds enqueued without waiting for any: the library's ceiling - this replay takes its challenges as inputs, "
                                               "a real prover cannot issue round k + 1 before it has round k's commitments",
                "synchronous_commitments": "the scope without SNARKVM_HIP_SCOPE_ASYNC_MSM: every commitment round a synchronous call"}
        legs = {name: dict(proof1_summary(sum(o[0]), o[0], o[2], None, len(mine)), mode=what[name]) for name, o in others.items()}
        print(json.dumps({
            "metric": "Varuna-proof-shaped hot-path replays per second, ONE proof at a time from one caller thread, commitments awaited in transcript order (BASELINE.json configs[3])",
            "value": world * len(mine) / dt_job,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": len(mine),
            "warmup": 4,
            "ms_per_step": dt_job / len(mine) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": "one proof at a time: 14 G1 commitments / openings of 2^16-2^17 pairs in 6 rounds, ~45 Fr NTTs of 2^16-2^18, polynomial passes, one 2^16 G2 MSM; "
                                   "device-resident random data, transfer_private domain sizes; one deferred-synchronisation scope per proof",
                       "proofs_per_rank": len(mine), "registered_srs": f"{ptab} tables x {pbits}-bit windows", "commitments": head, "mode": what[head], "pool_values": args.scalars,
                       "device_buffers": "snarkvm_hip_malloc + snarkvm_hip_memcpy_d2d / _memset through the C ABI (no torch on the data path)" if args.proof_mem == "hip" else "torch tensors"},
            "latency": proof1_summary(dt_rank, lat, times, grown, len(mine)),
            **legs,
            "g1_pairs_per_s": world * len(mine) * shape.pairs() / dt_job,
            "rank_ms_per_proof": [d / len(mine) * 1e3 for d in rank_dts],
            "rank_devices": rank_devices,
            "wall_ms_including_warmup": dt * 1e3,
            "checks": checks,
        }))
    keys.close()
    if world > 1:
        dist.destroy_process_group()


def proof_legs(dev_index, with_oracle):
    """Compact prover-shaped legs of the default line (rank 0, N = 1, outside the timed region, ~10 s): configs[3] (`proof1`), configs[4]
    on one GPU with 32 proofs in lock step (`proofs64`), and 8 caller threads issuing one proof-sized MSM per call (`concurrent_callers`:
    the reference's rayon fan-out, sonic_pc/mod.rs:186-245, met by the in-library coalescer).  Every leg carries its own `checks`."""
    import threading

    import torch

    from snarkvm_amd import _lib, proofs
    from snarkvm_amd.layout import G1_PROJECTIVE

    L = _lib.lib()
    shape = proofs.ProofShape()
    keys = proofs.ProverKeys(shape, tables=17, window_bits=15)
    out = {}
    P = 32
    salts = list(range(P))
    # ---- proof1: the headline is the order a prover is bound to (round k's commitments collected before round k + 1 is issued), in-stream
    dt_is, lat_is, got_is, times_is, grown = proof1_run(keys, dev_index, salts, await_rounds=True, msm_in_stream=True)
    dt_aw, lat_aw, got_aw, _, _ = proof1_run(keys, dev_index, salts, await_rounds=True)
    dt, lat, got1, times, _ = proof1_run(keys, dev_index, salts)
    leg = dict(proof1_summary(dt_is, lat_is, times_is, grown, P), value=P / dt_is, unit="proofs/s",
               commitments_awaited_on_further_streams={"ms_per_proof": dt_aw / P * 1e3, "value": P / dt_aw, "unit": "proofs/s",
                                                       "what": "the same order, the awaited rounds on further streams of the scope (an event hand-off per round)"},
               all_rounds_enqueued_at_once={"ms_per_proof": dt / P * 1e3, "value": P / dt, "unit": "proofs/s",
                                            "what": "the library's ceiling: all six commitment rounds enqueued without waiting for any (this replay takes its challenges as inputs; "
                                                    "no Fiat-Shamir prover can use this order) - round 5's headline"},
               what="one proof at a time from one caller thread, one SNARKVM_HIP_SCOPE_ASYNC_MSM scope per proof, the G2 MSM on a further stream, every commitment round "
                    "on the scope's own stream (SNARKVM_HIP_SCOPE_MSM_IN_STREAM) and collected (snarkvm_hip_scope_collect) before the next round is issued: the "
                    "Fiat-Shamir order of a real prover (bench.py --workload proof1)")
    if any([proofs.normalize_results(r) for r in g] != [proofs.normalize_results(r) for r in got1] for g in (got_aw, got_is)):
        raise SystemExit("bench.py: proof legs: awaiting the commitments round by round changes a result")
    out["proof1"] = leg
    # ---- the same call list through the reference's OWN three symbols on host buffers (what an unmodified snarkVM gets; bench.py --workload proof1 --ffi-only)
    ffi_salts = salts[:6]
    rows, res = proof1_ffi_rows(keys, ffi_salts, 4)
    for name, got in res.items():
        if [proofs.normalize_results(r) for r in got] != [proofs.normalize_results(r) for r in got1[: len(ffi_salts)]]:
            raise SystemExit(f"bench.py: proof legs: FFI row {name}: a result differs from the resident replay")
    out["proof1_ffi"] = {"value": rows["stateless"]["ms_inside_the_three_symbols_per_proof"], "unit": "ms/proof", "higher_is_better": False,
                         "stateless": rows["stateless"], "base_cache_16": rows["base_cache_16"],
                         "resident_ms_per_proof": dt_is / P * 1e3, "what": FFI_WHAT,
                         "checks": {"ffi_rows_vs_resident": f"all {len(ffi_salts)} proofs x 15 results of both rows == the resident replay (after affine normalisation)"}}
    # ---- proofs64 shape, 32 proofs in lock step (a scope per step, every commitment round one synchronous fused call), and the same group inside
    # ONE asynchronous scope (measured slower: the fused groups fill the chip by themselves)
    def lockstep(async_scope):
        lock = proofs.LockstepBatch(keys, group=P, devices=[dev_index], async_scope=async_scope)
        lock.run(salts)
        for ws in lock.workspaces:
            ws.times = {k: 0.0 for k in ws.times}
        _lib.check(L.snarkvm_hip_synchronize())
        t0 = time.perf_counter()
        _, got = lock.run(salts, collect=True)
        _lib.check(L.snarkvm_hip_synchronize())
        dt = time.perf_counter() - t0
        times = dict(lock.workspaces[0].times)
        del lock
        torch.cuda.empty_cache()
        return dt, got, times

    dt_lock, got_lock, t_lock = lockstep(False)
    dt_lock_async, got_lock_async, _ = lockstep(True)
    norm1 = [proofs.normalize_results(r) for r in got1]
    norm_lock = [proofs.normalize_results(r) for r in got_lock]
    if norm1 != norm_lock or norm_lock != [proofs.normalize_results(r) for r in got_lock_async]:
        raise SystemExit("bench.py: proof legs: the one-at-a-time replay and the lock-step replays differ")
    out["proofs64"] = {"value": P / dt_lock, "unit": "proofs/s", "proofs": P, "ms_per_proof": dt_lock / P * 1e3,
                       "g1_pairs_per_s_inside_msm_calls": P * shape.pairs() / t_lock["msm"] if t_lock.get("msm") else None,
                       "g2_pairs_per_s_inside_msm_calls": P * (1 << shape.lg_g2) / t_lock["g2"] if t_lock.get("g2") else None,
                       "call_time_ms_per_proof": {k: v / P * 1e3 for k, v in t_lock.items()},
                       "one_asynchronous_scope_per_group": {"value": P / dt_lock_async, "unit": "proofs/s", "ms_per_proof": dt_lock_async / P * 1e3},
                       "what": "BASELINE.json configs[4] on one GPU with 32 proofs in lock step (bench.py --workload proofs64 runs 64, and the concurrent-caller modes)",
                       "checks": {"lockstep_vs_proof1": f"all {P} proofs x 15 results identical in the lock-step replays and the one-at-a-time replay"}}
    leg["checks"] = {"proof1_vs_lockstep": f"all {P} proofs x 15 results identical in both replays"}
    # ---- checks (outside the timed regions)
    if with_oracle:
        p = salts[-1]
        secs = oracle_check_proof(keys, shape, got1[p], p)
        oracle_check_proof(keys, shape, got_lock[p], p)
        msg = f"proof {p}: all 14 G1 commitments / openings and the G2 MSM == oracle/proof_replay.py ({secs:.1f} s on the host)"
        leg["checks"]["one_proof_vs_oracle"] = msg
        pf = ffi_salts[-1]
        for name in res:
            oracle_check_proof(keys, shape, res[name][pf], pf)
        out["proof1_ffi"]["checks"]["one_proof_per_row_vs_oracle"] = f"proof {pf} of each row: all 14 G1 commitments / openings and the G2 MSM == oracle/proof_replay.py"
        out["proofs64"]["checks"]["one_proof_vs_oracle"] = msg
    # ---- the same two legs on witness-shaped data (shapes: synthesizer/process/src/tests/test_credits.rs:2868-2925): a pool whose VALUES are
    # 50 % zero / 25 % < 2^16 / 25 % uniform.  Only the 4 commitments that read the pool directly see that distribution (the other 10 commit to outputs of
    # transforms and divisions, which are uniform whatever went in) - which is also true of a real prover, whose commitments are all to coefficient-form polynomials.
    keys_w = proofs.ProverKeys(shape, tables=17, window_bits=15, scalars="witness")
    dt_w, lat_w, got_w, times_w, _ = proof1_run(keys_w, dev_index, salts, await_rounds=True, msm_in_stream=True)
    lock_w = proofs.LockstepBatch(keys_w, group=P, devices=[dev_index])
    lock_w.run(salts)
    _lib.check(L.snarkvm_hip_synchronize())
    t0 = time.perf_counter()
    _, got_lw = lock_w.run(salts, collect=True)
    _lib.check(L.snarkvm_hip_synchronize())
    dt_lw = time.perf_counter() - t0
    del lock_w
    torch.cuda.empty_cache()
    if [proofs.normalize_results(r) for r in got_w] != [proofs.normalize_results(r) for r in got_lw]:
        raise SystemExit("bench.py: proof legs (witness-like pool): the one-at-a-time replay and the lock-step replay differ")
    if with_oracle:
        oracle_check_proof(keys_w, shape, got_w[salts[-1]], salts[-1])
    wl_check = f"all {P} proofs x 15 results identical in the one-at-a-time and the lock-step replay" + (f"; proof {salts[-1]}: all 15 results == oracle/proof_replay.py" if with_oracle else "")
    out["proof1"]["witness_like"] = {"ms_per_proof": dt_w / P * 1e3, "value": P / dt_w, "unit": "proofs/s", "over_uniform": (P / dt_w) / out["proof1"]["value"], "checks": wl_check}
    out["proofs64"]["witness_like"] = {"ms_per_proof": dt_lw / P * 1e3, "value": P / dt_lw, "unit": "proofs/s", "over_uniform": (P / dt_lw) / out["proofs64"]["value"], "checks": wl_check}
    keys_w.close()
    # ---- concurrent callers: 8 threads, one MSM of 2^16 pairs per call, device-resident Montgomery scalars
    T, calls, n = 8, 24, 1 << shape.lg_r
    pool = torch.from_numpy(keys.pool_host.view(np.int64).reshape(-1)).cuda()
    torch.cuda.synchronize()

    def one_call(t, k, dst):
        ptr = ctypes.c_void_p(pool.data_ptr() + 32 * (97 * t + 13 * k))
        _lib.check(L.snarkvm_hip_msm_registered_ex(ctypes.c_void_p(dst.ctypes.data), keys.h, 0, n, 0, 0, ptr, 1, 1, 0))

    alone = np.zeros((T, calls), dtype=G1_PROJECTIVE)
    for t in range(T):
        for k in range(calls):
            one_call(t, k, alone[t, k : k + 1])
    t0 = time.perf_counter()
    for k in range(calls):
        one_call(0, k, alone[0, k : k + 1])
    dt_alone = time.perf_counter() - t0
    together = np.zeros((T, calls), dtype=G1_PROJECTIVE)
    L.snarkvm_hip_coalescer_stats(None, 1)
    start = threading.Barrier(T + 1)
    errors = []

    def worker(t):
        try:
            start.wait()
            for k in range(calls):
                one_call(t, k, together[t, k : k + 1])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    start.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    dt_thr = time.perf_counter() - t0
    if errors:
        raise SystemExit(f"bench.py: concurrent callers failed: {errors[:3]}")
    co = (ctypes.c_uint64 * 4)()
    L.snarkvm_hip_coalescer_stats(co, 0)
    from snarkvm_amd import kzg10
    if kzg10.to_affine(together.reshape(-1)).tobytes() != kzg10.to_affine(alone.reshape(-1)).tobytes():
        raise SystemExit("bench.py: concurrent callers: a coalesced result differs from the call issued alone")
    out["concurrent_callers"] = {"value": T * calls * n / dt_thr, "unit": "pairs/s", "caller_threads": T, "calls": T * calls, "pairs_per_call": n,
                                 "ms_per_call": dt_thr / calls * 1e3, "calls_per_s": T * calls / dt_thr,
                                 "one_thread_alone": {"value": calls * n / dt_alone, "unit": "pairs/s", "ms_per_call": dt_alone / calls * 1e3},
                                 "coalescer": {"batches": int(co[0]), "msm_instances": int(co[1]), "largest_batch": int(co[2]), "single_instance_batches": int(co[3]),
                                               "instances_per_batch": (co[1] / co[0]) if co[0] else None},
                                 "what": "8 caller threads, each issuing synchronous snarkvm_hip_msm_registered_ex calls of 2^16 pairs (one commitment per call, the reference's "
                                         "rayon fan-out); calls that meet inside the library travel as fused groups",
                                 "checks": {"coalesced_vs_alone": f"all {T * calls} results identical (after to_affine) to the same call issued alone"}}
    del pool
    keys.close()
    torch.cuda.empty_cache()
    return out


def proofs64(args, rank, world, dev_index, barrier, max_over_ranks, gather_over_ranks, rank_devices):
    """BASELINE.json configs[4]: a batch of Varuna-proof-shaped call lists, sharded over the ranks (64 / N proofs each).  Inside a
    rank two ways of issuing them are timed, one after the other, on the same proofs:
      lockstep   `VarunaSNARK::prove_batch` is a batch by construction (varuna.rs:336): round k of all proofs of a group is ONE
                 fused MSM call, the transforms of a step ONE batched NTT call per size (snarkvm_amd/proofs.py::LockstepBatch);
      callers    --proof-workers concurrent caller threads, one proof each at a time - the reference's rayon fan-out; their
                 proof-sized MSMs meet in the library's coalescer.
    `value` is the lock-step rate.  Every result of both modes is compared (after affine normalisation) and, on rank 0, all 15
    results of one proof are checked against the CPU oracle's restatement of the same data flow (oracle/proof_replay.py)."""
    import torch
    import torch.distributed as dist

    from snarkvm_amd import _lib, proofs

    shape = proofs.ProofShape()
    ptab, pbits = (int(v) for v in args.proof_geometry.split("x"))
    keys = proofs.ProverKeys(shape, tables=ptab, window_bits=pbits, scalars=args.scalars)
    mine = list(range(rank, args.proofs, world))
    checks = {}
    # ---- lock step
    lock = proofs.LockstepBatch(keys, group=min(args.proof_group, max(1, len(mine))), devices=[dev_index], async_scope=args.lockstep_async)
    lock.run(mine[: lock.group])  # warm-up: allocations, twiddle tables
    for ws in lock.workspaces:
        ws.times = {k: 0.0 for k in ws.times}
    barrier()
    t0 = time.perf_counter()
    _, got_lock = lock.run(mine, collect=True)
    barrier()
    dt_lock_rank = time.perf_counter() - t0
    dt_lock = max_over_ranks(dt_lock_rank)
    rank_dts = gather_over_ranks(dt_lock_rank)
    print(f"[bench rank {rank}] lock-step: {len(mine)} proofs in {dt_lock_rank * 1e3:.1f} ms on this rank", file=sys.stderr)
    t_lock = dict(lock.workspaces[0].times)
    del lock
    torch.cuda.empty_cache()
    # ---- concurrent callers: (a) every caller thread issues its proof inside one asynchronous scope (replay_single), (b) one synchronous
    # call per step (replay: the proof-sized MSMs of concurrent callers meet in the coalescer) - round 4's mode, kept beside it
    L = _lib.lib()

    def callers(scope, async_msm=True):
        batch = proofs.ProofBatch(keys, workers=args.proof_workers, devices=[dev_index], scope=scope, async_msm=async_msm)
        batch.run(list(range(len(batch.workspaces))))  # warm-up: one proof per worker
        for ws in batch.workspaces:
            ws.times = {k: 0.0 for k in ws.times}
        barrier()
        L.snarkvm_hip_coalescer_stats(None, 1)
        t0 = time.perf_counter()
        _, got = batch.run(mine, collect=True)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        co = (ctypes.c_uint64 * 4)()
        L.snarkvm_hip_coalescer_stats(co, 0)
        times = {k: sum(ws.times[k] for ws in batch.workspaces) for k in batch.workspaces[0].times}
        n_ws = len(batch.workspaces)
        del batch
        torch.cuda.empty_cache()
        return dt, got, co, times, n_ws

    # (a) a scope per proof for the transforms and passes + synchronous commitment rounds that meet other callers' rounds in the coalescer (the
    # fastest: `concurrent_callers.value`); (b) everything of a proof in one asynchronous scope; (c) round 4's mode, one synchronous call per step
    dt_thr, got_thr, co_mix, t_scope, n_workers = callers(True, async_msm=False)
    dt_mix, got_mix, _, _, _ = callers(True, async_msm=True)
    dt_ser, got_ser, co, t_thr, _ = callers(False)
    # ---- checks (outside the timed regions)
    norm_lock = [proofs.normalize_results(r) for r in got_lock]
    norm_thr = [proofs.normalize_results(r) for r in got_thr]
    norm_ser = [proofs.normalize_results(r) for r in got_ser]
    norm_mix = [proofs.normalize_results(r) for r in got_mix]
    for i, p in enumerate(mine):
        if norm_lock[i] != norm_thr[i] or norm_lock[i] != norm_ser[i] or norm_lock[i] != norm_mix[i]:
            raise SystemExit(f"bench.py: proof {p}: the lock-step replay and the concurrent-caller replays differ")
    checks["lockstep_vs_callers"] = f"all {len(mine)} proofs of this rank: 14 commitments + the G2 result identical in the lock-step replay and both caller modes"
    if rank == 0 and mine and not args.no_cpu_baseline:
        from oracle import cpu as oracle
        from oracle import proof_replay

        oracle.set_threads(min(os.cpu_count() or 1, 64))
        p = mine[-1]
        t0 = time.perf_counter()
        want = proof_replay.expected_results(keys.pool_host, keys.g1_host, keys.g2_host, keys.point, shape.lg_r, shape.lg_k, shape.lg_g2, shape.nmax, p)
        oracle_s = time.perf_counter() - t0
        got = got_lock[mine.index(p)]
        from snarkvm_amd import kzg10
        from snarkvm_amd.layout import G1_PROJECTIVE, G2_PROJECTIVE
        for j in range(14):
            ga = kzg10.to_affine(np.frombuffer(got[j], dtype=G1_PROJECTIVE))
            wa = want[j]
            if not (np.array_equal(ga["x"], wa["x"]) and np.array_equal(ga["y"], wa["y"]) and np.array_equal(ga["infinity"], wa["infinity"])):
                raise SystemExit(f"bench.py: proof {p}, commitment {j}: the device result differs from the CPU oracle")
        if want[14] is not None:
            g2a = oracle.g2_to_affine(np.frombuffer(got[14], dtype=G2_PROJECTIVE))
            if g2a.tobytes() != want[14].tobytes():
                raise SystemExit(f"bench.py: proof {p}: the G2 MSM result differs from the CPU oracle")
        checks["one_proof_vs_oracle"] = (f"proof {p}: all 14 G1 commitments / openings and the G2 MSM == oracle/proof_replay.py (fft_in_place, pointwise passes, "
                                         f"divide_with_q_and_r, batched::msm, standard::msm restatements; {oracle_s:.1f} s on the host)")
    if rank == 0:
        print(json.dumps({
            "metric": "Varuna-proof-shaped hot-path replays per second (BASELINE.json configs[4]: batch of 64, G1 + G2)",
            "value": args.proofs / dt_lock,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.proofs,
            "warmup": min(args.proof_group, len(mine)),
            "ms_per_step": dt_lock / args.proofs * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": "64 x (14 G1 commitments / openings of 2^16-2^17 pairs in 6 rounds, ~45 Fr NTTs of 2^16-2^18, polynomial passes, one 2^16 G2 MSM); "
                                   "device-resident random data, transfer_private domain sizes; lock-step batch (prove_batch shape)",
                       "proofs": args.proofs, "lockstep_group": min(args.proof_group, max(1, len(mine))), "proofs_per_rank": len(mine),
                       "registered_srs": f"{ptab} tables x {pbits}-bit windows", "pool_values": args.scalars,
                       "lockstep_form": "one SNARKVM_HIP_SCOPE_ASYNC_MSM scope per group" if args.lockstep_async else "a scope per step, synchronous commitment calls"},
            "g1_pairs_per_s": args.proofs * shape.pairs() / dt_lock,
            "g2_pairs_per_s": args.proofs * (1 << shape.lg_g2) / dt_lock,
            "rank0_call_time_ms_per_proof": {k: v / max(1, len(mine)) * 1e3 for k, v in t_lock.items()},
            # the fused G1 rate on the proof mix: pairs of this rank / wall time spent inside its snarkvm_hip_msm_registered_batch_ex calls
            # (not with --lockstep-async: an asynchronous scope's calls return at once)
            "g1_pairs_per_s_inside_msm_calls": (len(mine) * shape.pairs() / t_lock["msm"]) if (t_lock.get("msm") and not args.lockstep_async) else None,
            "g2_pairs_per_s_inside_msm_calls": (len(mine) * (1 << shape.lg_g2) / t_lock["g2"]) if (t_lock.get("g2") and not args.lockstep_async) else None,
            "concurrent_callers": {"value": args.proofs / dt_thr, "unit": "proofs/s", "ms_per_proof": dt_thr / args.proofs * 1e3,
                                   "caller_threads_per_rank": n_workers, "g1_pairs_per_s": args.proofs * shape.pairs() / dt_thr,
                                   "coalescer": {"batches": int(co_mix[0]), "msm_instances": int(co_mix[1]), "largest_batch": int(co_mix[2]), "single_instance_batches": int(co_mix[3]),
                                                 "instances_per_batch": (co_mix[1] / co_mix[0]) if co_mix[0] else None},
                                   "rank0_host_time_ms_per_proof": {k: v / max(1, len(mine)) * 1e3 for k, v in t_scope.items()},
                                   "what": "one proof per caller thread at a time (the reference's rayon fan-out): a deferred-synchronisation scope per proof for the transforms and "
                                           "passes (no wait per call, a round's independent transforms batched), the commitment rounds as synchronous calls that meet other callers' "
                                           "rounds in the in-library coalescer (snarkvm_amd/proofs.py::replay_single, async_msm=False)",
                                   "one_asynchronous_scope_per_proof": {
                                       "value": args.proofs / dt_mix, "unit": "proofs/s", "ms_per_proof": dt_mix / args.proofs * 1e3,
                                       "what": "SNARKVM_HIP_SCOPE_ASYNC_MSM | _STABLE_INPUTS per proof (the single-caller latency path, bench.py --workload proof1): nothing is fused across callers"},
                                   "one_synchronous_call_per_step": {
                                       "value": args.proofs / dt_ser, "unit": "proofs/s", "ms_per_proof": dt_ser / args.proofs * 1e3,
                                       "coalescer": {"batches": int(co[0]), "msm_instances": int(co[1]), "largest_batch": int(co[2]), "single_instance_batches": int(co[3]),
                                                     "instances_per_batch": (co[1] / co[0]) if co[0] else None},
                                       "rank0_call_time_ms_per_proof": {k: v / max(1, len(mine)) * 1e3 for k, v in t_thr.items()},
                                       "what": "round 4's caller mode: every step a synchronous call; proof-sized MSMs of concurrent callers fused by the in-library coalescer"}},
            "rank_ms_per_proof": [d / max(1, len(mine)) * 1e3 for d in rank_dts],
            "rank_devices": rank_devices,
            "checks": checks,
        }))
    keys.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
