#!/usr/bin/env python3
"""Multi-GPU readiness on a one-GPU box: the reference's multi-GPU MSM (algorithms/cuda/cuda/snarkvm.cu:254-295: point-range
split over the devices, host combine) run through the reference's own symbol `snarkvm_msm` at 2^24 with one, two, four and
eight LOGICAL devices (the same GPU listed k times: own streams, workspaces, staging buffers and uploader thread per logical
device).  Asserts that every device set returns the identical affine point (== the closed form), and records the per-device
chunk timelines (`SNARKVM_HIP_TRACE=1`: host timestamps of upload / compute per chunk) so that a run on a real 8-GPU node can
be checked against them.

  python tools/multidevice_split.py [lg=24] > gpurun_out/multidevice_split.md

Each device set is its own subprocess (the device set is fixed at the first compute call of a process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import torch
from snarkvm_amd import _lib, msm, plugin, synthetic
from snarkvm_amd.layout import G1_AFFINE, G1_PROJECTIVE
lg = %(lg)d
n = 1 << lg
L = _lib.lib()
torch.cuda.set_device(0)
buf = torch.empty(n * G1_AFFINE.itemsize, dtype=torch.uint8, device="cuda")
_lib.check(L.snarkvm_hip_g1_generate_bases_device(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(1), ctypes.c_size_t(n)))
bases = buf.cpu().numpy().view(G1_AFFINE)
del buf
torch.cuda.empty_cache()
sc = synthetic.random_fr_integers(n, 0xD15C)
sys.stderr.write("TRACE_BEGIN warm\n")
got = plugin.msm(bases, sc)
sys.stderr.write("TRACE_BEGIN timed\n")
t0 = time.perf_counter()
got = plugin.msm(bases, sc)
dt = time.perf_counter() - t0
sys.stderr.write("TRACE_END\n")
aff = np.zeros(1, dtype=G1_AFFINE)
_lib.check(L.snarkvm_hip_g1_to_affine(ctypes.c_void_p(aff.ctypes.data), ctypes.c_void_p(got.ctypes.data), ctypes.c_size_t(1)))
print("RESULT", msm.num_devices(), "%%.3f" %% (dt * 1e3), aff.tobytes().hex())
'''


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    results = {}
    print(f"# `snarkvm_msm` at 2^{lg} (host bases + host scalars, stateless) split over k logical devices of ONE MI355X\n")
    print("The k device threads share one GPU and one PCIe link here, so the wall time cannot drop with k; what the run shows is that")
    print("the split, the per-device chunk rings and the host combine produce the identical point, and what each device's timeline")
    print("looks like.  On a node every logical device below is a physical GPU with its own link.\n")
    traces = {}
    for devs in ("0", "0,0", "0,0,0,0", "0,0,0,0,0,0,0,0"):
        env = dict(os.environ, SNARKVM_HIP_DEVICES=devs, SNARKVM_HIP_TRACE="1")
        env.pop("SNARKVM_HIP_BASE_CACHE", None)
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "lg": lg}], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
        if r.returncode != 0:
            print("FAILED", devs, r.stdout[-2000:], r.stderr[-4000:])
            sys.exit(1)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
        results[devs] = (int(line[1]), float(line[2]), line[3])
        err = r.stderr.split("TRACE_BEGIN timed\n")[-1].split("TRACE_END")[0]
        traces[devs] = "".join(l + "\n" for l in err.splitlines() if "amdgpu.ids" not in l)
    ref = results["0"][2]
    print("| SNARKVM_HIP_DEVICES | logical devices | call ms | affine result |")
    print("|---|---|---|---|")
    for devs, (nd, ms, hx) in results.items():
        print(f"| {devs} | {nd} | {ms:.2f} | {'identical to the one-device result' if hx == ref else 'DIFFERENT'} ({hx[:16]}...) |")
    assert all(v[2] == ref for v in results.values()), "device sets disagree"
    for devs, t in traces.items():
        print(f"\n## per-chunk timeline, SNARKVM_HIP_DEVICES={devs} (ms since the call began)\n\n```\n{t}```")
    print("\nMULTIDEVICE_SPLIT_OK")


if __name__ == "__main__":
    main()
