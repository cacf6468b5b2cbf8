"""Runs in its own process with SNARKVM_HIP_NO_TORCH=1 (tests/test_gpu_devmem.py): one proof's call list (snarkvm_amd/proofs.py::replay_single) with EVERY
device buffer allocated, filled and copied through the C ABI (snarkvm_hip_malloc / _memcpy_h2d / _memcpy_d2d / _memset) - what a Rust host without a HIP
crate does.  torch must never be imported here; the results go to an .npz the parent compares with the oracle and with the torch-backed replay."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from snarkvm_amd import _lib, proofs  # noqa: E402


def note(msg):
    print(f"[torch_free_replay] {msg}", file=sys.stderr, flush=True)


def main(out_path, seed, salts):
    import faulthandler

    faulthandler.enable()
    assert os.environ.get("SNARKVM_HIP_NO_TORCH") == "1"
    note(f"devices visible: {_lib.device_count()}")
    shape = proofs.ProofShape(lg_r=12, lg_k=13, lg_g2=10)
    keys = proofs.ProverKeys(shape, seed=seed, mem="hip")
    note("keys registered")
    ws = proofs.SingleProofWorkspace(keys)
    assert ws.mem == "hip"
    note("workspace allocated")
    stats = np.zeros(5, dtype=np.uint64)
    out = {}
    for salt in salts:
        for name, kw in (("async", dict(async_msm=True)), ("sync", dict(async_msm=False)), ("await", dict(async_msm=True, await_rounds=True)),
                         ("in_stream", dict(async_msm=True, await_rounds=True, msm_in_stream=True))):
            got = []
            proofs.replay_single(ws, salt, got, **kw)
            note(f"proof {salt} mode {name} done")
            assert len(got) == 15
            out[f"{name}_{salt}_g1"] = np.frombuffer(b"".join(got[:14]), dtype=np.uint8)
            out[f"{name}_{salt}_g2"] = np.frombuffer(got[14], dtype=np.uint8)
    # a second pass of a warmed shape allocates nothing inside the library (the caller's own HipMem blocks are not counted)
    _lib.lib().snarkvm_hip_alloc_stats(None, 1)
    proofs.replay_single(ws, salts[0], [], async_msm=True, await_rounds=True, msm_in_stream=True)
    _lib.lib().snarkvm_hip_alloc_stats(ctypes.c_void_p(stats.ctypes.data), 0)  # (a bare Python int would travel as a 32-bit C int)
    out["alloc_stats"] = stats
    out["num_devices"] = np.array([_lib.lib().snarkvm_hip_num_devices()])
    assert "torch" not in sys.modules, "torch was imported on the torch-free path"
    np.savez(out_path, **out)
    ws.pool.free()
    ws.work.free()
    keys.close()
    note("buffers released")
    print("OK torch-free replay:", len(salts), "proofs x 4 modes", flush=True)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), [int(s) for s in sys.argv[3].split(",")])
