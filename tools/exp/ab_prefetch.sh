set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03c; mkdir -p $O
A="--steps 6 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout 300 python bench.py $A > $O/bench_prefetch.json 2> $O/bench_prefetch.err
for s in 96 128; do SNARKVM_HIP_TUNING=seg=$s timeout 300 python bench.py $A > $O/bench_S$s.json 2> $O/bench_S$s.err; done
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
timeout 200 python tools/bench_round.py > $O/bench_round.txt 2>&1
for f in $O/bench_prefetch.json $O/bench_S96.json $O/bench_S128.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], "ms/step", round(d["ms_per_step"],3), "value %.4g"%d["value"], d["phase_ms"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
tail -n 3 $O/pytest_gpu.log; grep -v amdgpu $O/bench_round.txt | tail -1
