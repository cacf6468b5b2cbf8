# rocprofv3 kernel statistics of the fused 14-commitment batch call alone (20 timed + 2 warm-up calls)
set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=gpurun_out/r03r; mkdir -p $O
BENCH_ROUND_MODES=all14 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python tools/bench_round.py > $O/run.log 2>&1
D=$(find $O/prof -name "*.db" | head -1); [ -n "$D" ] && python tools/rocprof_summary.py stats $D > $O/r03_rocprofv3_kernel_stats_fused_round.txt
find $O -name "*.db" -delete; grep -v amdgpu $O/run.log | tail -1 | cut -c1-200; head -30 $O/r03_rocprofv3_kernel_stats_fused_round.txt | cut -c1-130
