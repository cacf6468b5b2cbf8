#!/usr/bin/env python3
"""bench.py - BLS12-377 G1 MSM (pairs/s) + Fr NTT (elements/s) on MI355X.

One "step" = one G1 variable-base MSM of 2^lg_msm scalar-point pairs (BASELINE.json configs[1]: 2^24 by default)
with bases registered in HBM and scalars resident in HBM when the timed region starts.  `value` = pairs/s over all
ranks (each rank runs its own independent MSM instance: instance-level sharding, no data-path collective -> weak
scaling).  The JSON line also carries the NTT throughput at 2^24 (the second half of BASELINE.json's metric), a
`roofline` object for the dominant kernel and for the scalar-read phase, and a `cpu_baseline` (the C++ restatement of
the reference's rayon path, oracle/, timed on this box's host cores on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
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
    ap.add_argument("--cpu-lg-msm", type=int, default=23)
    ap.add_argument("--cpu-lg-ntt", type=int, default=24)
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
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)

    from snarkvm_amd import _lib, synthetic
    from snarkvm_amd.layout import G1_AFFINE
    from snarkvm_amd.msm import RegisteredBases

    L = _lib.lib()
    _lib.check(L.snarkvm_hip_set_device(ctypes.c_int(dev_index)))

    def barrier():
        torch.cuda.synchronize()
        _lib.check(L.snarkvm_hip_synchronize())
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
    rb = RegisteredBases(device_ptr=bases_dev.data_ptr(), npoints=n, tables=args.tables,
                         window_bits=0 if (args.table_bits == 16 and args.tables == 16) or args.tables == 1 else args.table_bits)
    del bases_dev
    scalars = synthetic.random_fr_integers(n, synthetic.SEED_MSM_LARGE + rank)
    d_scalars = torch.from_numpy(scalars.view(np.int64)).cuda()
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
    t0 = time.perf_counter()
    if args.no_pipeline:
        for _ in range(args.steps):
            res = rb.msm(device_ptr=d_scalars.data_ptr(), npoints=n, window_bits=args.window_bits)
    else:
        # K independent MSM instances (a batch of commitments) pipelined over the backend's HIP streams: the
        # latency-bound tail of one instance overlaps the accumulation of the next.  Every step does the full work.
        res = rb.msm_batch(device_ptrs=[d_scalars.data_ptr()] * args.steps, npoints=[n] * args.steps, window_bits=args.window_bits)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    pairs_per_s = world * n * args.steps / dt

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

    def ntt(direction):
        _lib.check(L.snarkvm_hip_ntt_device(ctypes.c_void_p(d_x.data_ptr()), ctypes.c_uint32(args.lg_ntt), 0, direction, 0))

    ntt(0)
    ntt(1)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.ntt_steps):
        ntt(i & 1)
    barrier()
    ntt_dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([ntt_dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ntt_dt = float(t.item())
    ntt_elems_per_s = world * nn * args.ntt_steps / ntt_dt
    L.snarkvm_hip_set_profiling(1)
    ntt(0)
    ntt_kernel_ms = L.snarkvm_hip_get_phase_ms(0)
    L.snarkvm_hip_set_profiling(0)

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu as oracle

        cores = os.cpu_count() or 1
        threads = min(cores, 64)
        oracle.set_threads(threads)
        cn = 1 << args.cpu_lg_msm
        gen = np.zeros(1, dtype=oracle.G1_AFFINE)
        gen["x"] = [1171681672315280277, 6528257384425852712, 7514971432460253787, 2032708395764262463, 12876543207309632302, 107509843840671767]
        gen["y"] = [13572190014569192121, 15344828677741220784, 17067903700058808083, 10342263224753415805, 1083990386877464092, 21335464879237822]
        cn = min(cn, n)
        cb = oracle.g1_gen_bases(gen, 1, cn)
        cs = scalars[:cn]
        oracle.g1_msm(cb[:1024], cs[:1024])  # warm-up
        t0 = time.perf_counter()
        cpu_res = oracle.g1_msm(cb, cs, oracle.MSM_BATCHED)
        cpu_msm_dt = time.perf_counter() - t0
        cnn = min(1 << args.cpu_lg_ntt, nn)
        cx = x[:cnn].copy()
        t0 = time.perf_counter()
        oracle.ntt(cx)
        cpu_ntt_dt = time.perf_counter() - t0
        cpu = {
            "value": cn / cpu_msm_dt,
            "unit": "pairs/s",
            "cores": threads,
            "kind": "port",
            "sample": f"batched::msm restatement (oracle/cpu_oracle.cpp, OpenMP one task per window) on {cn} pairs of the "
                      f"same workload: {cpu_msm_dt:.2f} s; fft_in_place restatement on {cnn} elements: {cpu_ntt_dt:.2f} s",
            "host_cores": cores,
            "ntt_value": cnn / cpu_ntt_dt,
            "ntt_unit": "elements/s",
        }

    # HBM bytes per dispatch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs; see
    # profiles/r01_pmc_traffic.json for provenance and the gfx950 x2 FETCH correction).  Only quoted for the
    # configuration they were collected on; PMC collection cannot run inside this process.
    pmc = {}
    try:
        if args.lg_msm == 24 and args.lg_ntt == 24 and args.tables == 12 and args.table_bits == 22 and not args.window_bits:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)["kernels"]
    except Exception:
        pmc = {}

    valu = {}
    try:
        if pmc:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_valu.json")) as f:
                valu = json.load(f)["kernels"]
    except Exception:
        valu = {}

    def valu_issue(kernel, ms, launches=1):
        """VALU issue utilisation: PMC wave-instruction count (committed, same configuration) x the 4-cycle wave64 issue
        floor of a 16-lane SIMD / (1024 SIMDs x 2.4 GHz x measured kernel time)."""
        k = valu.get(kernel)
        if not k or not ms:
            return None
        cycles = ms * 1e-3 * 2.4e9 * 1024
        cpi = cycles / (k["SQ_INSTS_VALU"] * launches)
        return {"wave_insts_per_launch": k["SQ_INSTS_VALU"], "cycles_per_inst": cpi, "issue_floor_cycles": 4.0, "frac": 4.0 / cpi}

    def traffic(kernel, fetch_key, times=1):
        k = pmc.get(kernel)
        return None if not k else (k[fetch_key] + k["write_bytes"]) * times

    if rank == 0:
        acc_ms = phase_ms.get("msm_accumulate", 0.0)
        dig_ms = phase_ms.get("msm_digits", 0.0)
        cbits = args.window_bits or (args.table_bits if args.tables > 1 else 16)
        W = args.tables * (args.table_bits // cbits) if args.tables > 1 else (254 + cbits - 1) // cbits  # digit rows per scalar
        out = {
            "metric": "BLS12-377 G1 MSM scalar-point pairs/sec (+ Fr NTT elements/sec in ntt_*)",
            "value": pairs_per_s,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (29-bit radix) modular integer arithmetic, Fq 377-bit / Fr 253-bit",
            "data": "synthetic",
            "config": {"workload": f"G1 Pippenger MSM 2^{args.lg_msm} (BASELINE.json configs[1]), bases (i+1)G registered in HBM, "
                                   f"uniform scalars in HBM; independent instance per GPU",
                       "lg_msm": args.lg_msm, "lg_ntt": args.lg_ntt, "window_bits": args.window_bits or "auto", "base_tables": args.tables, "table_bits": args.table_bits,
                       "pipelined_batch": not args.no_pipeline},
            "ntt_value": ntt_elems_per_s,
            "ntt_unit": "elements/s",
            "ntt_ms_per_transform": ntt_dt / args.ntt_steps * 1e3,
            "ntt_kernel_ms": ntt_kernel_ms,
            "phase_ms": {k: round(v, 4) for k, v in phase_ms.items()},
            # dominant kernel: bucket accumulation.  It is ALU-bound; its algorithmic HBM bytes are the gathered
            # bases (96 B) + sorted index (4 B) per (pair, window) - reported against the HBM peak for context.
            "roofline": {
                "bound": "hbm",
                "kernel": "msm_accumulate_seg_kernel",
                "achieved": (n * 100.0 * W) / (acc_ms * 1e-3) / 1e9 if acc_ms else None,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": ((n * 100.0 * W) / (acc_ms * 1e-3) / 1e9 / 8000.0) if acc_ms else None,
                "traffic": traffic("msm_accumulate_seg_kernel<Fp<FqP>, 1>", "fetch_bytes_raw"),
                "algorithmic_bytes": n * 100.0 * W,
                "note": "whole-MSM is integer-ALU bound (SURVEY.md 8d): see valu_issue; roofline_scalar_read is the HBM-bound phase",
                "valu_issue": valu_issue("msm_accumulate_seg_kernel<Fp<FqP>, 1>", acc_ms),
            },
            # the phase north_star scopes the HBM claim to: scalar read + digit extraction, 32 B per scalar
            "roofline_scalar_read": {
                "bound": "hbm",
                "kernel": "msm_digits_kernel",
                "achieved": (32.0 * n) / (dig_ms * 1e-3) / 1e9 if dig_ms else None,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": ((32.0 * n) / (dig_ms * 1e-3) / 1e9 / 8000.0) if dig_ms else None,
                "traffic": traffic("msm_digits_kernel<unsigned int>", "fetch_bytes_x2"),
                "algorithmic_bytes": 32.0 * n,
                "bytes_incl_digit_writes_GBps": ((32.0 + (4.0 if cbits > 16 else 2.0) * W) * n) / (dig_ms * 1e-3) / 1e9 if dig_ms else None,
            },
            "roofline_ntt": {
                "bound": "hbm",
                "kernel": "ntt_pass_kernel_v2 (3 passes)",
                "achieved": (64.0 * nn) / (ntt_kernel_ms * 1e-3) / 1e9 if ntt_kernel_ms else None,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": ((64.0 * nn) / (ntt_kernel_ms * 1e-3) / 1e9 / 8000.0) if ntt_kernel_ms else None,
                "traffic": traffic("ntt_pass_kernel_v2", "fetch_bytes_x2", times=3),
                "valu_issue": valu_issue("ntt_pass_kernel_v2", ntt_kernel_ms, launches=3),
                "algorithmic_bytes": 64.0 * nn,
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
