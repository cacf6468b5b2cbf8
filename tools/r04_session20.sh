#!/bin/bash
# Round 4, GPU session 20: one reduce round of 64 instead of two rounds of 8 (existing knobs reduce_rounds / seg2).
O=gpurun_out/r04_s20; mkdir -p $O
export TMPDIR=/tmp
for v in "reduce_rounds=2" "reduce_rounds=1,seg2=64" "reduce_rounds=1,seg2=16" "reduce_rounds=1,seg2=8"; do
  SNARKVM_HIP_TUNING=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --ntt-steps 2 > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], "ms/step", round(d["ms_per_step"], 2), {k: round(v, 3) for k, v in d["phase_ms"].items() if "acc" in k or "reduce" in k})
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done
