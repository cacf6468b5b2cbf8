set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03i; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python tools/bench_g2.py > $O/g2.md 2>&1
timeout 300 python tools/phase_profile.py 14 16 17 20 24 > $O/phases.md 2>&1
timeout 200 python tools/bench_round.py > $O/bench_round.txt 2>&1
timeout 300 python bench.py --workload proofs64 --proof-workers 8 > $O/proofs64.json 2> $O/proofs64.err
tail -n 3 $O/pytest_gpu.log
python - <<'PY'
import json
for f in ("bench","proofs64"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03i/{f}.json") if l.startswith("{")][-1])
        print(f, "%.4g"%d["value"], round(d["ms_per_step"],3), d.get("ntt_value"), d.get("phase_ms"), d.get("tables1_ms_per_step"), (d.get("end_to_end_ffi") or {}).get("snarkvm_msm_2p24",{}).get("call_ms_samples"))
    except Exception as e: print(f, "ERR", e)
PY
grep -v amdgpu $O/g2.md | cut -c1-420; grep -v amdgpu $O/phases.md | grep -A3 "^###" | cut -c1-200; grep -v amdgpu $O/bench_round.txt | tail -1 | cut -c1-400
