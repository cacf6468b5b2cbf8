#!/usr/bin/env python3
"""The lock-step half of `bench.py --workload proofs64` alone (no concurrent-caller mode, no result checks), for a kernel trace that
shows what one lock-step group is made of:
  rocprofv3 --kernel-trace --stats -d gpurun_out/lockprof -o l -- python tools/profile_lockstep.py [proofs] [group]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from snarkvm_amd import proofs  # noqa: E402


def main():
    nproofs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    group = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    torch.cuda.set_device(0)
    shape = proofs.ProofShape()
    keys = proofs.ProverKeys(shape)
    lock = proofs.LockstepBatch(keys, group=group, devices=[0])
    lock.run(list(range(group)))
    for ws in lock.workspaces:
        ws.times = {k: 0.0 for k in ws.times}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lock.run(list(range(nproofs)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"lock step: {nproofs} proofs in {dt * 1e3:.1f} ms = {dt / nproofs * 1e3:.2f} ms per proof; call time per proof (ms):",
          {k: round(v / nproofs * 1e3, 3) for k, v in lock.workspaces[0].times.items()})
    keys.close()


if __name__ == "__main__":
    main()
