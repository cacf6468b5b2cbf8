#!/bin/bash
# Round 4, GPU session 19: kernel trace of the lock-step replay on ONE lane (no overlap between groups: true kernel durations).
O=gpurun_out/r04_s19; mkdir -p $O
export TMPDIR=/tmp
SNARKVM_HIP_TUNING=lanes=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/lock -o l -- python tools/profile_lockstep.py 64 32 > $O/lockstep.log 2>&1; grep "lock step" $O/lockstep.log
D=$(find $O/lock -name "*.db" | head -1); [ -n "$D" ] && python tools/rocprof_summary.py stats $D > $O/r04_rocprofv3_kernel_stats_lockstep_one_lane.txt; find $O -name "*.db" -delete; head -24 $O/r04_rocprofv3_kernel_stats_lockstep_one_lane.txt | cut -c1-125
