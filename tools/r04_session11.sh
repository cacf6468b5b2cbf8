#!/bin/bash
# Round 4, GPU session 11: one bucket sink for the scalar chunks of a host-scalar MSM over registered bases; ramp / taper parity.
O=gpurun_out/r04_s11; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multidevice.py -x -q --timeout 500 -k "ramp or chunk_ring or two_logical" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in "taper=1" "taper=0" "taper=1,scalar_chunk_lg=21" "taper=1,scalar_chunk_lg=23"; do
  SNARKVM_HIP_TUNING=$v timeout 200 python tools/reg_host_scalars.py 20 22 24 2> "$O/reg_$v.err" | tee -a $O/reg_host_scalars.md
done
SNARKVM_HIP_TRACE=1 timeout 200 python tools/reg_host_scalars.py 24 > /dev/null 2> $O/reg_trace_2p24.err; grep -c "uploaded" $O/reg_trace_2p24.err
SNARKVM_HIP_BASE_CACHE=0 timeout 150 python tools/ffi_msm_sweep.py 16 18 20 22 24 2> "$O/ffi_default.err" | tee -a $O/ffi_sweep.md
