#!/bin/bash
# Round 4, GPU session 17: kernel trace of the lock-step replay alone; the GPU suite under eight logical devices.
O=gpurun_out/r04_s17; mkdir -p $O
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/lock -o l -- python tools/profile_lockstep.py 64 32 > $O/lockstep.log 2>&1; tail -2 $O/lockstep.log
D=$(find $O/lock -name "*.db" | head -1); [ -n "$D" ] && python tools/rocprof_summary.py stats $D > $O/r04_rocprofv3_kernel_stats_lockstep.txt; find $O -name "*.db" -delete; head -30 $O/r04_rocprofv3_kernel_stats_lockstep.txt | cut -c1-125
SNARKVM_HIP_DEVICES=0,0,0,0,0,0,0,0 timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "not 2_24 and not 2_22 and not 2_25 and not bench_ and not ramp" > $O/r04_pytest_gpu_eight_logical_devices.log 2>&1; echo "eight-device suite rc=$?"; tail -3 $O/r04_pytest_gpu_eight_logical_devices.log
