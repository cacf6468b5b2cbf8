#!/bin/bash
# Round 4, GPU session 8: the C++ concurrent-callers bench (coalescer on / off), the GPU suite under two logical devices.
O=gpurun_out/r04_s8; mkdir -p $O
export TMPDIR=/tmp
g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_callers.cpp -o /tmp/bench_callers -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib || exit 1
for v in "coalesce=1" "coalesce=0" "coalesce=1,coalesce_us=150"; do echo "== $v"; GPU_MAX_HW_QUEUES=8 SNARKVM_HIP_TUNING=$v timeout 120 /tmp/bench_callers 1 2 4 8 16 32 2> "$O/callers_$v.err" | tee "$O/callers_$v.md"; done
SNARKVM_HIP_DEVICES=0,0 timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "not 2_24 and not 2_22 and not 2_25 and not bench_" > $O/r04_pytest_gpu_two_logical_devices.log 2>&1; echo "two-device suite rc=$?"; tail -3 $O/r04_pytest_gpu_two_logical_devices.log
