#!/bin/bash
# Round 4, GPU session 22: kernel trace of the C++ concurrent-caller proof replay (8 threads): how busy is the GPU, what does a proof cost there?
O=gpurun_out/r04_s22; mkdir -p $O
export TMPDIR=/tmp
g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_proof_callers.cpp -o /tmp/bench_proof_callers -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib || exit 1
GPU_MAX_HW_QUEUES=8 timeout 200 rocprofv3 --kernel-trace --stats -d $O/cal -o c -- /tmp/bench_proof_callers - 8 > $O/callers.log 2>&1; grep "^| 8\|one caller" $O/callers.log
D=$(find $O/cal -name "*.db" | head -1); [ -n "$D" ] && python tools/rocprof_summary.py stats $D > $O/r04_rocprofv3_kernel_stats_proof_callers.txt; find $O -name "*.db" -delete; head -16 $O/r04_rocprofv3_kernel_stats_proof_callers.txt | cut -c1-125
python - <<'PY'
import re
tot = 0.0
for l in open("gpurun_out/r04_s22/r04_rocprofv3_kernel_stats_proof_callers.txt").read().split("\n")[1:]:
    m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", l)
    if m and "precompute_table" not in m.group(1) and "generate_bases" not in m.group(1): tot += float(m.group(3))
print("kernel time besides registration, summed over streams: %.1f ms for 64 serial + 8 warm-up + 64 concurrent proofs (no G2 leg)" % (tot / 1e3))
PY
