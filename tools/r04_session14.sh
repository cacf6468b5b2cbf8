#!/bin/bash
# Round 4, GPU session 14: 2-D grids of the tile-counter kernels (level-1 scan); 2^20 / 2^22 table geometries.
O=gpurun_out/r04_s14; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --timeout 200 -k "wide or fused or 2_20 or tables or registered" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --ntt-steps 2 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_s14/bench.json"))
print("ms/step", round(d["ms_per_step"], 2), "whole phase", round(d["roofline_scalar_read"]["whole_phase"]["ms"], 3), {k: round(v, 3) for k, v in d["phase_ms"].items() if "sort" in k or "scalar" in k})
PY
for g in "16 16" "15 17" "14 19" "13 20"; do set -- $g
  timeout 200 python bench.py --lg-msm 20 --lg-ntt 20 --tables $1 --table-bits $2 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --ntt-steps 2 > $O/b20_$1x$2.json 2> $O/b20_$1x$2.err
  python - $O/b20_$1x$2.json "2^20 $1x$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), "pairs/s", f"{d['value']:.3e}", {k: round(v, 3) for k, v in d["phase_ms"].items()})
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done
