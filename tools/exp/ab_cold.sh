# A/B on one box: cold paths out of line (default build) vs everything inlined (-DSV_COLD_INLINE build in libsnarkvm_hip_inline.so)
set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03j; mkdir -p $O
for v in default inline default inline; do
  if [ $v = inline ]; then export SNARKVM_HIP_LIB=$PWD/snarkvm_amd/lib/libsnarkvm_hip_inline.so; else unset SNARKVM_HIP_LIB; fi
  echo "== $v" >> $O/phases.md; timeout 300 python tools/phase_profile.py 14 16 17 20 24 2>&1 | grep -v amdgpu | grep -A3 "^###" | grep -v "^--" >> $O/phases.md
  echo "== $v" >> $O/g2.md; timeout 300 python tools/bench_g2.py 2>&1 | grep "^| 1[268]" | cut -c1-330 >> $O/g2.md
done
cat $O/phases.md | grep -v "^|---\|msm_digits\|msm_scalar_read"; cat $O/g2.md
unset SNARKVM_HIP_LIB
A="--steps 6 --warmup 1 --no-cpu-baseline --no-extra-legs"
for r in 2 1 0; do SNARKVM_HIP_TUNING=reduce_rounds=$r timeout 300 python bench.py $A > $O/rr$r.json 2> $O/rr$r.err; python - "$O/rr$r.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], "ms/step", round(d["ms_per_step"],3), {k:round(v,2) for k,v in d["phase_ms"].items()})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
