#!/bin/bash
# The evidence run of the round on the GPU box: full GPU suite, the default bench line, rocprofv3 kernel stats + PMC passes
# (tools/profile_round.sh), side benches.  Outputs: gpurun_out/r03f/ and gpurun_out/r03prof/.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r03f
mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/r03_pytest_gpu.log 2>&1; echo rc=$? >> $O/r03_pytest_gpu.log)
timeout 400 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1
timeout 200 python tools/bench_round.py > $O/bench_round.txt 2>&1
timeout 300 python tools/bench_g2.py > $O/g2.md 2>&1
timeout 120 python tools/ntt_small.py > $O/ntt_small.md 2>&1
timeout 300 python tools/phase_profile.py 14 16 17 18 20 22 24 > $O/phases.md 2>&1
timeout 300 python bench.py --workload proofs64 --proof-workers 8 > $O/proofs64.json 2> $O/proofs64.err
timeout 300 python bench.py --workload proofs64 --proof-workers 1 > $O/proofs64_1caller.json 2> $O/proofs64_1caller.err
tail -n 3 $O/r03_pytest_gpu.log
python - <<'PY'
import json
for f in ("r03_bench_default","proofs64","proofs64_1caller"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03f/{f}.json") if l.startswith("{")][-1])
        print(f, "%.4g"%d["value"], round(d["ms_per_step"],3), d.get("ntt_value"), d.get("ntt_ms_per_transform"), (d.get("alu_roofline") or {}).get("frac"), (d.get("alu_roofline") or {}).get("mad_frac"))
    except Exception as e: print(f, "ERR", e)
PY
