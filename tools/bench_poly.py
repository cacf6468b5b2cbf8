#!/usr/bin/env python3
"""Prover-round Fr vector kernels and G1 (de)serialisation on one MI355X, device-resident operands, next to the CPU
oracle (oracle/liboracle.so, the restated reference functions) on the host cores.  Prints a markdown table (committed
under profiles/).  Effective bandwidth = algorithmic bytes (every input read once + every output written once) / time."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import cpu as oracle  # noqa: E402  (CPU baseline leg only)
from snarkvm_amd import _lib, serialize, synthetic  # noqa: E402
from snarkvm_amd.layout import G1_AFFINE  # noqa: E402


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    L = _lib.lib()
    torch.cuda.set_device(0)
    oracle.set_threads(min(64, oracle.max_threads()))
    rows = []
    nmax = 1 << 24
    host = oracle.fr_op("from_bigint", synthetic.random_fr_integers(nmax, 7))
    d = [torch.from_numpy(np.roll(host, k, axis=0).view(np.int64)).cuda() for k in range(3)]
    d_out = torch.empty_like(d[0])
    scalar = host[5:6].copy()
    sp = ctypes.c_void_p(scalar.ctypes.data)
    rem = np.zeros((1, 4), dtype=np.uint64)
    torch.cuda.synchronize()
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    for lg in (16, 20, 24):
        n = 1 << lg
        reps = 50 if lg <= 20 else 10
        cpu_n = min(n, 1 << 20)
        h = [x[:cpu_n] for x in (host, np.roll(host, 1, axis=0), np.roll(host, 2, axis=0))]

        def gpu(name, fn, bytes_per_elem, cpu_fn=None):
            t = timed(fn, reps)
            cpu = ""
            if cpu_fn is not None:
                t0 = time.perf_counter()
                cpu_fn()
                cpu = f"{cpu_n / (time.perf_counter() - t0):.3e}"
            rows.append(f"| {name} | {lg} | {t * 1e3:.4f} | {n / t:.3e} | {bytes_per_elem * n / t / 1e9:.0f} | {cpu} |")

        gpu("a*b-c (rowcheck)", lambda: _lib.check(L.snarkvm_hip_fr_vec_op(3, P(d_out), P(d[0]), P(d[1]), P(d[2]), None, ctypes.c_size_t(n), 1)), 128,
            lambda: oracle.fr_vec_op("mul_sub", h[0], h[1], h[2]))
        gpu("p / (X - z) + p(z)", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_linear(P(d_out), ctypes.c_void_p(rem.ctypes.data), P(d[0]), ctypes.c_size_t(n), sp, 1)), 64,
            lambda: oracle.poly_divide(h[0], [(0, scalar[0]), (1, host[9])]))
        gpu("p(z) only", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_linear(None, ctypes.c_void_p(rem.ctypes.data), P(d[0]), ctypes.c_size_t(n), sp, 1)), 32,
            lambda: oracle.poly_evaluate(h[0], scalar))
        d_out.copy_(d[0])
        gpu("batch_inversion_and_mul", lambda: _lib.check(L.snarkvm_hip_fr_batch_inversion_and_mul(P(d_out), ctypes.c_size_t(n), sp, 1)), 64,
            lambda: oracle.batch_inversion_and_mul(h[0], scalar))
        gpu("distribute_powers", lambda: _lib.check(L.snarkvm_hip_fr_distribute_powers(P(d_out), ctypes.c_size_t(n), sp, sp, 1)), 64,
            lambda: oracle.distribute_powers(h[0], scalar, scalar))
        gpu("lagrange coefficients", lambda: _lib.check(L.snarkvm_hip_fr_lagrange_coefficients(P(d_out), ctypes.c_uint32(lg), sp, 1)), 32,
            (lambda: oracle.lagrange_coefficients(min(lg, 20), scalar)))
        gpu("p / (X^(n/2) - 1)", lambda: _lib.check(L.snarkvm_hip_fr_divide_by_vanishing(P(d_out), P(d[2]), P(d[0]), ctypes.c_size_t(n), ctypes.c_size_t(n // 2), 1)), 64)
    # (de)serialisation: host bytes in, registered bases out (includes the PCIe copy of 96 / 48 B per point)
    for lg in (16, 20):
        n = 1 << lg
        buf = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
        _lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(n)))
        aff = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=G1_AFFINE)
        for comp in (False, True):
            data = serialize.g1_serialize(aff, compressed=comp)
            serialize.g1_deserialize(data, compressed=comp)  # warm-up: staging buffers are allocated on first use
            t0 = time.perf_counter()
            back = serialize.g1_deserialize(data, compressed=comp)
            t = time.perf_counter() - t0
            assert np.array_equal(back["x"], aff["x"]) and np.array_equal(back["y"], aff["y"])
            rows.append(f"| G1 deserialize {'compressed' if comp else 'uncompressed'} (host bytes -> host G1Affine) | {lg} | {t * 1e3:.3f} | {n / t:.3e} | | |")
            L.snarkvm_hip_free_bases(serialize.register_bases_serialized(data, n, comp, False, 1))
            t0 = time.perf_counter()
            h_ = serialize.register_bases_serialized(data, n, comp, False, 1)
            t = time.perf_counter() - t0
            L.snarkvm_hip_free_bases(h_)
            rows.append(f"| SRS bytes -> registered bases ({'compressed' if comp else 'uncompressed'}) | {lg} | {t * 1e3:.3f} | {n / t:.3e} | | |")
    # setup-time group operations (N4): host buffers in and out
    from snarkvm_amd import group

    gen = np.zeros(1, dtype=G1_AFFINE)
    gen["x"] = [1171681672315280277, 6528257384425852712, 7514971432460253787, 2032708395764262463, 12876543207309632302, 107509843840671767]
    gen["y"] = [13572190014569192121, 15344828677741220784, 17067903700058808083, 10342263224753415805, 1083990386877464092, 21335464879237822]
    for lg in (16, 20):
        n = 1 << lg
        tab = group.FixedBase.get_window_table(253, group.FixedBase.get_mul_window_size(n), gen)
        group.FixedBase.msm(253, 0, tab, host[:1024])
        t0 = time.perf_counter()
        group.FixedBase.msm(253, 0, tab, host[:n])
        t = time.perf_counter() - t0
        cn = 1 << 14
        t0 = time.perf_counter()
        oracle.g1_fixed_base_msm(gen, host[:cn])
        ct = time.perf_counter() - t0
        rows.append(f"| FixedBase::msm (host buffers) | {lg} | {t * 1e3:.3f} | {n / t:.3e} | | {cn / ct:.3e} (2^14 sample) |")
    for lg in (12, 16):
        n = 1 << lg
        proj = group.FixedBase.msm(253, 0, tab, host[:n])
        group.group_ntt(proj[:16], inverse=True)
        t0 = time.perf_counter()
        group.group_ntt(proj, inverse=True)
        t = time.perf_counter() - t0
        cn = 1 << 10
        t0 = time.perf_counter()
        oracle.g1_group_ntt(proj[:cn], inverse=True)
        ct = time.perf_counter() - t0
        rows.append(f"| group-element iFFT (lagrange_basis, host buffers) | {lg} | {t * 1e3:.3f} | {n / t:.3e} | | {cn / ct:.3e} (2^10 sample) |")
    print("| kernel | lg n | ms | elements/s | effective GB/s | CPU oracle elements/s (<= 2^20 sample) |")
    print("|---|---|---|---|---|---|")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
