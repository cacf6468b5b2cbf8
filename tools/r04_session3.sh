#!/bin/bash
# Round 4, GPU session 3: snarkvm_msm over host buffers with the bucket sink + tapered chunks (A/B against taper=0), parity of the chunked paths.
O=gpurun_out/r04_s3; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_parity.py -x -q -k "chunk or ffi or split or stateless or config1 or (ab_switches and taper)" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in "taper=1" "taper=0" "taper=1,msm_chunk_lg=20" "taper=1,msm_chunk_lg=22"; do
  echo "== $v"; SNARKVM_HIP_TUNING=$v SNARKVM_HIP_BASE_CACHE=0 timeout 300 python tools/bench_ffi.py 16 18 20 22 24 2> "$O/ffi_$v.err" | tee "$O/ffi_$v.md" | head -9
done
SNARKVM_HIP_TRACE=1 SNARKVM_HIP_BASE_CACHE=0 timeout 300 python tools/bench_ffi.py 24 > $O/ffi_trace.md 2> $O/ffi_trace.err; grep "snarkvm_hip" $O/ffi_trace.err | tail -30
