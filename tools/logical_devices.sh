#!/bin/bash
# The GPU suite on 2 and 8 LOGICAL devices of one GPU (SNARKVM_HIP_DEVICES=0,0[,0...]: independent streams, workspaces and base replicas per
# entry - every host-operand MSM then takes the split / replicated paths, batches are dealt over the devices).  The >= 2^22 cases and the
# bench subprocesses are left out (they size workspaces for one device).  gpurun -- 'bash tools/logical_devices.sh r05'
TAG=${1:-r05}; O=gpurun_out/${TAG}_logical; mkdir -p $O; export TMPDIR=/tmp
SNARKVM_HIP_DEVICES=0,0 timeout 1200 python -m pytest tests -m gpu -q -x --timeout 300 -k "not 2_24 and not 2_22 and not 2_25 and not bench_" > $O/${TAG}_pytest_gpu_two_logical_devices.log 2>&1; echo "two-device suite rc=$?"; tail -3 $O/${TAG}_pytest_gpu_two_logical_devices.log
SNARKVM_HIP_DEVICES=0,0,0,0,0,0,0,0 timeout 1200 python -m pytest tests -m gpu -q -x --timeout 300 -k "not 2_24 and not 2_22 and not 2_25 and not bench_ and not ramp" > $O/${TAG}_pytest_gpu_eight_logical_devices.log 2>&1; echo "eight-device suite rc=$?"; tail -3 $O/${TAG}_pytest_gpu_eight_logical_devices.log
