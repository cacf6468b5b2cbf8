set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03l; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
timeout 300 python tools/bench_g2.py 2>&1 | grep "^| 1[268]" | cut -c1-330 > $O/g2.md
timeout 300 python tools/phase_profile.py 14 16 17 20 24 2>&1 | grep -v amdgpu | grep -A3 "^###" | grep -v "^--" > $O/phases.md
timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --workload proofs64 --proof-workers 8 > $O/proofs64.json 2> $O/proofs64.err
tail -n 3 $O/pytest_gpu.log; cat $O/g2.md; grep -v "^|---\|msm_digits\|msm_scalar_read" $O/phases.md
python - <<'PY'
import json
for f in ("bench","proofs64"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03l/{f}.json") if l.startswith("{")][-1]); print(f, "%.4g"%d["value"], round(d["ms_per_step"],3), d.get("phase_ms"))
    except Exception as e: print(f,"ERR",e)
PY
