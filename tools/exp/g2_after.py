"""Where in a transcript-ordered proof should the independent G2 MSM be issued?  replay_single(g2_after=k) for k = 0 .. 5, 32 proofs each."""
import sys, time
sys.path.insert(0, '/root/repo')
from snarkvm_amd import _lib, proofs
_lib.check(_lib.lib().snarkvm_hip_set_device(0))
keys = proofs.ProverKeys(proofs.ProofShape(), tables=17, window_bits=15, mem="hip")
ws = proofs.SingleProofWorkspace(keys)
ref = None
for k in (0, 1, 2, 3, 4, 5, 0):
    for s in range(4):
        proofs.replay_single(ws, s, None, True, None, True, True, k)
    lat = []
    got = []
    for s in range(32):
        t0 = time.perf_counter(); proofs.replay_single(ws, s, got if s == 7 else None, True, None, True, True, k); lat.append(time.perf_counter() - t0)
    n = proofs.normalize_results(got)
    ref = ref or n
    assert n == ref
    lat.sort()
    print(f"g2_after={k}: mean {sum(lat) / len(lat) * 1e3:.3f} ms, median {lat[16] * 1e3:.3f}, min {lat[0] * 1e3:.3f}", flush=True)
