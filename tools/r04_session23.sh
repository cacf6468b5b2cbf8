#!/bin/bash
# Round 4, GPU session 23: fold variant and reduce group size at 2^24 (tuning only, no rebuild).
O=gpurun_out/r04_s23; mkdir -p $O
export TMPDIR=/tmp
for v in "fold_flat=1" "fold_flat=0" "seg2=32" "seg2=12"; do
  SNARKVM_HIP_TUNING=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --ntt-steps 2 > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], "ms/step", round(d["ms_per_step"], 2), {k: round(v, 3) for k, v in d["phase_ms"].items() if "acc" in k or "reduce" in k})
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done
