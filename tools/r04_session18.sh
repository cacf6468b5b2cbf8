#!/bin/bash
# Round 4, GPU session 18: the G1 tail on the lazy arithmetic (tuning lazy_tail) - parity, A/B at 2^24, proofs64.
O=gpurun_out/r04_s18; mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_proofs.py -x -q --timeout 200 -k "not 2_24 and not 2_25 and not two_rank" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 400 python -m pytest tests/test_gpu_multidevice.py -x -q --timeout 300 -k "ramp or chunk_ring or two_logical or lazy_tail" >> $O/pytest.log 2>&1; echo "pytest2 rc=$?"; tail -2 $O/pytest.log
for v in "lazy_tail=1" "lazy_tail=0"; do
  SNARKVM_HIP_TUNING=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --ntt-steps 2 > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], "ms/step", round(d["ms_per_step"], 2), {k: round(v, 3) for k, v in d["phase_ms"].items() if "acc" in k or "reduce" in k or "finish" in k})
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
for v in "lazy_tail=1" "lazy_tail=0"; do
  SNARKVM_HIP_TUNING=$v timeout 200 python bench.py --workload proofs64 --no-cpu-baseline > $O/p64_$v.json 2> $O/p64_$v.err
  python - $O/p64_$v.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["concurrent_callers"]
    print(f"{sys.argv[2]} lockstep {d['value']:.1f}/s ({d['ms_per_step']:.2f} ms) {({k: round(v, 2) for k, v in d['rank0_call_time_ms_per_proof'].items()})} | callers {c['value']:.1f}/s")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
