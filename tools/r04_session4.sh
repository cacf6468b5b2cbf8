#!/bin/bash
# Round 4, GPU session 4: signed-limb NTT butterflies (parity + A/B), the geometry sweep at 2^23 / 2^24 (cost of doubling the buckets),
# the default bench with the shader-clock sampler.
O=gpurun_out/r04_s4; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_proofs.py -x -q -k "ntt or kat or polymul or domain or proofs or lockstep" > $O/pytest_ntt.log 2>&1; echo "pytest_ntt rc=$?"; tail -3 $O/pytest_ntt.log
for v in "ntt_signed=1" "ntt_signed=0"; do
  SNARKVM_HIP_TUNING=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline --ntt-steps 20 > "$O/bench_$v.json" 2> "$O/bench_$v.err"
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{sys.argv[2]:16s} ntt 2^24 {d['ntt_ms_per_transform']:.3f} ms/transform ({d['ntt_value']:.3e} el/s) kernels {d['ntt_kernel_ms']:.3f} ms  sync-call {d['ntt_sync_call']['ms_per_transform']:.3f}  | msm {d['ms_per_step']:.2f} ms/step sclk {d['alu_roofline'].get('sclk_during_timed_steps')} mad_frac {d['alu_roofline']['mad_frac']:.3f} -> {d['alu_roofline'].get('mad_frac_at_sustained_clock')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
for v in "ntt_signed=1" "ntt_signed=0"; do echo "== $v"; SNARKVM_HIP_TUNING=$v timeout 200 python tools/ntt_small.py 2>&1 | tail -12; done
timeout 900 python tools/geometry_sweep.py 23 24 2> $O/geometry.err | tee $O/geometry_23_24.md
