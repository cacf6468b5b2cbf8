set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03d; mkdir -p $O
A="--steps 8 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout 300 python bench.py $A > $O/base.json 2> $O/base.err
SNARKVM_HIP_TUNING=acc_one_wg=1 timeout 300 python bench.py $A > $O/onewg96.json 2> $O/onewg96.err
SNARKVM_HIP_TUNING=acc_one_wg=1,acc_lds=83968 timeout 300 python bench.py $A > $O/onewg82.json 2> $O/onewg82.err
SNARKVM_HIP_TUNING=acc_one_wg=1,acc_lds=83968,lanes=4 timeout 300 python bench.py $A > $O/onewg82_l4.json 2> $O/onewg82_l4.err
SNARKVM_HIP_TUNING=acc_one_wg=1,acc_lds=83968 timeout 300 python bench.py $A --no-pipeline > $O/onewg82_sync.json 2> $O/onewg82_sync.err
for f in base onewg96 onewg82 onewg82_l4 onewg82_sync; do python - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], "ms/step", round(d["ms_per_step"],3), "value %.4g"%d["value"], {k:round(v,2) for k,v in d["phase_ms"].items()})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
