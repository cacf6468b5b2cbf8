#!/bin/bash
# Round 4, GPU session 12: geometric scalar chunks of a host-scalar MSM over registered bases (parity + sweep).
O=gpurun_out/r04_s12; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multidevice.py -x -q --timeout 500 -k "ramp or chunk_ring" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in "scalar_geo=4" "scalar_geo=0" "scalar_geo=3" "scalar_geo=6" "scalar_geo=8" "taper=0"; do
  SNARKVM_HIP_TUNING=$v timeout 200 python tools/reg_host_scalars.py 20 22 24 2> "$O/reg_$v.err" | tee -a $O/reg_host_scalars.md
done
SNARKVM_HIP_TRACE=1 timeout 200 python tools/reg_host_scalars.py 24 > /dev/null 2> $O/reg_trace_2p24.err; grep "snarkvm_hip" $O/reg_trace_2p24.err | tail -8
