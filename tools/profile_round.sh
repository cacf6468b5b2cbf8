#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace statistics of bench.py and three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters;
# separate runs with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Run on the GPU box from the repo
# root: bash tools/profile_round.sh <tag>; outputs go to gpurun_out/<tag>prof/.
set -u
TAG=${1:-r05}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=gpurun_out/${TAG}prof
mkdir -p $OUT
ARGS1="--steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-pipeline"
ARGS0="--steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-extra-legs --ntt-steps 2"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG -- python bench.py $ARGS1 > $OUT/bench_under_rocprof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- python bench.py $ARGS0 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- python bench.py $ARGS0 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d $OUT/sq -o s -- python bench.py $ARGS0 > $OUT/pmc_sq.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/sq2 -o t -- python bench.py $ARGS0 > $OUT/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/proofs -o p -- python bench.py --workload proofs64 --no-cpu-baseline > $OUT/proofs64_under_rocprof.log 2>&1
find $OUT -name "*.db" | head
S=$(find $OUT/stats -name "*.db" | head -1); F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1); Q=$(find $OUT/sq -name "*.db" | head -1)
[ -n "$S" ] && python tools/rocprof_summary.py schema $S > $OUT/rocpd_schema.txt
[ -n "$S" ] && python tools/rocprof_summary.py stats $S > $OUT/${TAG}_rocprofv3_kernel_stats.txt
P=$(find $OUT/proofs -name "*.db" | head -1)
[ -n "$P" ] && python tools/rocprof_summary.py stats $P > $OUT/${TAG}_rocprofv3_kernel_stats_proofs64.txt
[ -n "$F" ] && python tools/rocprof_summary.py pmc $F > $OUT/${TAG}_rocprofv3_pmc_fetch.txt
[ -n "$W" ] && python tools/rocprof_summary.py pmc $W > $OUT/${TAG}_rocprofv3_pmc_write.txt
[ -n "$Q" ] && python tools/rocprof_summary.py pmc $Q > $OUT/${TAG}_rocprofv3_pmc_sq_counters.txt
Q2=$(find $OUT/sq2 -name "*.db" | head -1)
[ -n "$Q2" ] && python tools/rocprof_summary.py pmc $Q2 > $OUT/${TAG}_rocprofv3_pmc_sq_stall_counters.txt
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on 'bench.py $ARGS0' (MSM 2^24 over 12 base tables of 22-bit windows, fused scalar read; NTT 2^24), MI355X, $TAG" > $OUT/${TAG}_pmc_traffic.json
# keep the merge small: the sqlite files stay on the box
find $OUT -name "*.db" -delete
ls -la $OUT
