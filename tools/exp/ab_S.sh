# segment length sweep at 2^24 (default 128 from 2^27 entries on): longer segments leave fewer partial sums for the tail
set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03p; mkdir -p $O
A="--steps 6 --warmup 1 --no-cpu-baseline --no-extra-legs"
for s in 128 192 256 384 128 256; do SNARKVM_HIP_TUNING=seg=$s timeout 300 python bench.py $A > $O/S$s.json 2> $O/S$s.err; python - "$O/S$s.json" $s <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("S", sys.argv[2], "ms/step", round(d["ms_per_step"],3), {k[4:]:round(v,2) for k,v in d["phase_ms"].items()})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
