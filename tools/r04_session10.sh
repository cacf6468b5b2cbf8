#!/bin/bash
# Round 4, GPU session 10: front ramp of the snarkvm_msm chunk ring (parity + sweep), proof geometry 17x15 vs 16x16 in lock step.
O=gpurun_out/r04_s10; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_multidevice.py -x -q --timeout 400 -k "ramp=3 or chunk_ring" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in "ramp=0" "ramp=2" "ramp=3" "ramp=4" "ramp=3,msm_chunk_lg=20"; do
  SNARKVM_HIP_BASE_CACHE=0 SNARKVM_HIP_TUNING=$v timeout 150 python tools/ffi_msm_sweep.py 20 22 24 2> "$O/ffi_$v.err" | tee -a $O/ffi_sweep.md
done
SNARKVM_HIP_TRACE=1 SNARKVM_HIP_BASE_CACHE=0 timeout 150 python tools/ffi_msm_sweep.py 24 > /dev/null 2> $O/ffi_trace_2p24.err; grep -c "uploaded" $O/ffi_trace_2p24.err
for g in 17x15 16x16; do
  timeout 200 python bench.py --workload proofs64 --no-cpu-baseline --proof-geometry $g > "$O/p64_$g.json" 2> "$O/p64_$g.err"
  python - "$O/p64_$g.json" "$g" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["concurrent_callers"]
    print(f"{sys.argv[2]:24s} lockstep {d['value']:.1f}/s ({d['ms_per_step']:.2f} ms) {({k: round(v, 2) for k, v in d['rank0_call_time_ms_per_proof'].items()})} g1 {d.get('g1_pairs_per_s_inside_msm_calls', 0):.3e} | callers {c['value']:.1f}/s")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
