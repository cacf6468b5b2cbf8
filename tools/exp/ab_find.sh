set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03m; mkdir -p $O
for v in prev new prev new; do
  if [ $v = prev ]; then export SNARKVM_HIP_LIB=$PWD/snarkvm_amd/lib/libsnarkvm_hip_prev.so; else unset SNARKVM_HIP_LIB; fi
  echo "== $v" >> $O/phases.md; timeout 300 python tools/phase_profile.py 14 16 17 20 24 2>&1 | grep -v amdgpu | grep -A3 "^###" | grep -v "^--\|^|---\|msm_digits\|msm_scalar_read" >> $O/phases.md
  echo "== $v" >> $O/round.md; timeout 200 python tools/bench_round.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-330 >> $O/round.md
done
unset SNARKVM_HIP_LIB
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sonic.py -q -x -k "msm or sonic or kzg or varuna" > $O/pytest_msm.log 2>&1; echo rc=$? >> $O/pytest_msm.log)
cat $O/phases.md $O/round.md; tail -n 3 $O/pytest_msm.log
