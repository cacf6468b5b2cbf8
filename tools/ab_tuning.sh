#!/bin/bash
# Same-box A/B of SNARKVM_HIP_TUNING settings (replaces the one-shot tools/r04_session*.sh of round 4).
#   gpurun -- 'bash tools/ab_tuning.sh <tag> [--proofs64|--proof1] [--bench-args "..."] -- "<tuning A>" "<tuning B>" ...'
# Each setting runs bench.py once; one line per setting goes to stdout and the JSON lines to gpurun_out/<tag>/.
# An empty string "" is the default build.  Example (round 4, session 20):
#   bash tools/ab_tuning.sh r05_reduce -- "reduce_rounds=2" "reduce_rounds=1,seg2=64"
TAG=$1; shift
WORKLOAD=msm; ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --ntt-steps 2"
while [ "$1" != "--" ] && [ $# -gt 0 ]; do
  case "$1" in
    --proofs64) WORKLOAD=proofs64; ARGS="--no-cpu-baseline";;
    --proof1) WORKLOAD=proof1; ARGS="--no-cpu-baseline";;
    --bench-args) shift; ARGS="$1";;
  esac; shift
done
shift
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1)); f=$O/bench_$i
  SNARKVM_HIP_TUNING="$v" timeout 600 python bench.py --workload $WORKLOAD $ARGS > $f.json 2> $f.err
  python - $f.json "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    keep = {k: round(v, 3) for k, v in d.get("phase_ms", {}).items()}
    print(f"[{sys.argv[2] or 'default'}] {d['metric']}: {d['value']:.4e} {d['unit']}, {d['ms_per_step']:.3f} ms/step", keep)
except Exception as e:
    print(f"[{sys.argv[2]}] FAILED {e}"); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
