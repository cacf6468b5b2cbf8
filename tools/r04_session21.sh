#!/bin/bash
# Round 4, GPU session 21: the concurrent-caller proof replay without an interpreter (tools/bench_proof_callers.cpp), coalescer on / off.
O=gpurun_out/r04_s21; mkdir -p $O
export TMPDIR=/tmp
g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_proof_callers.cpp -o /tmp/bench_proof_callers -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib || exit 1
timeout 100 python -c "
import sys; sys.path.insert(0, '.')
from snarkvm_amd import synthetic
synthetic.g2_points(1 << 16).tofile('/tmp/g2pts_65536.bin')" || exit 1
for v in "coalesce=1" "coalesce=0"; do echo "== $v"; GPU_MAX_HW_QUEUES=8 SNARKVM_HIP_TUNING=$v timeout 150 /tmp/bench_proof_callers /tmp/g2pts_65536.bin 1 4 8 16 2> "$O/proof_callers_$v.err" | tee "$O/proof_callers_$v.md"; tail -3 "$O/proof_callers_$v.err"; done
