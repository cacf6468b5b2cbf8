set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03e; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_poly.py -q -x -k "ntt or abi or poly or varuna" > $O/pytest_ntt.log 2>&1; echo rc=$? >> $O/pytest_ntt.log)
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload proofs64 --proof-workers 8 > $O/proofs64.json 2> $O/proofs64.err
tail -n 3 $O/pytest_ntt.log
python - <<'PY'
import json
for f in ("bench_default","proofs64"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03e/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d.get("ntt_value"), d.get("ntt_ms_per_transform"), d.get("ntt_sync_call"), d.get("ntt_2p20"), d.get("rank0_call_time_ms_per_proof"))
    except Exception as e: print(f, "ERR", e)
PY
