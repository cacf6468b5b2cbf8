#!/bin/bash
# Round 4, GPU session 15: level-2/3 histograms with private copies, unrolled scan loops - parity + phase times.
O=gpurun_out/r04_s15; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q --timeout 200 -k "msm and not 2_24 and not 2_25" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --ntt-steps 2 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_s15/bench.json"))
print("ms/step", round(d["ms_per_step"], 2), "whole phase", round(d["roofline_scalar_read"]["whole_phase"]["ms"], 3), {k: round(v, 3) for k, v in d["phase_ms"].items()})
PY
timeout 200 python bench.py --workload proofs64 --no-cpu-baseline > $O/p64.json 2> $O/p64.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_s15/p64.json")); c = d["concurrent_callers"]
print(f"lockstep {d['value']:.1f}/s ({d['ms_per_step']:.2f} ms) {({k: round(v, 2) for k, v in d['rank0_call_time_ms_per_proof'].items()})} | callers {c['value']:.1f}/s")
PY
