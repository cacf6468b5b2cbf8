# A/B on one box: cold paths out of line (default build) vs everything inlined (-DSV_COLD_INLINE build in libsnarkvm_hip_inline.so)
set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03j; mkdir -p $O
for v in default inline default inline; do
  if [ $v = inline ]; then export SNARKVM_HIP_LIB=$PWD/snarkvm_amd/lib/libsnarkvm_hip_inline.so; else unset SNARKVM_HIP_LIB; fi
  echo "== $v" >> $O/phases.md; timeout 300 python tools/phase_profile.py 14 16 17 20 24 2>&1 | grep -v amdgpu | grep -A3 "^###" | grep -v "^--" >> $O/phases.md
  echo "== $v" >> $O/g2.md; timeout 300 python tools/bench_g2.py 2>&1 | grep "^| 1[268]" | cut -c1-330 >> $O/g2.md
done
cat $O/phases.md | grep -v "^|---\|msm_digits\|msm_scalar_read"; cat $O/g2.md
