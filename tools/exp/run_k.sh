set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03k; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_poly.py tests/test_gpu_sonic.py tests/test_gpu_parity.py -q -x -k "poly or sonic or kzg or varuna or divide or open" > $O/pytest_poly.log 2>&1; echo rc=$? >> $O/pytest_poly.log)
timeout 300 python tools/bench_poly.py > $O/bench_poly.md 2>&1
timeout 300 python bench.py --workload proofs64 --proof-workers 8 > $O/proofs64.json 2> $O/proofs64.err
timeout 300 python bench.py --workload proofs64 --proof-workers 1 > $O/proofs64_1.json 2> $O/proofs64_1.err
tail -n 3 $O/pytest_poly.log; grep -v amdgpu $O/bench_poly.md | tail -25 | cut -c1-200
python - <<'PY'
import json
for f in ("proofs64","proofs64_1"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03k/{f}.json") if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d.get("rank0_call_time_ms_per_proof"))
    except Exception as e: print(f,"ERR",e)
PY
