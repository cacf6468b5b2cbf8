#!/bin/bash
# Round 4, GPU session 7: lazy G2 with the exceptional path inlined - diagnostic, parity, A/B, proofs64.  Every step under a short timeout.
O=gpurun_out/r04_s7; mkdir -p $O
export TMPDIR=/tmp
timeout 60 python tools/exp/g2lazy_diag.py > $O/diag.log 2>&1; echo "diag rc=$?"; tail -4 $O/diag.log
grep -q DIAG_DONE $O/diag.log || { echo "diagnostic did not finish: stopping"; exit 1; }
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multidevice.py tests/test_gpu_proofs.py -x -q -k "g2 or G2 or proofs or lockstep or chunk_ring or two_logical" > $O/pytest_g2.log 2>&1; echo "pytest_g2 rc=$?"; tail -3 $O/pytest_g2.log
for v in "lazy2=1" "lazy2=0"; do echo "== $v"; SNARKVM_HIP_TUNING=$v timeout 120 python tools/bench_g2.py 2> "$O/g2_$v.err" | tee "$O/g2_$v.md" | tail -4; done
for v in "lazy2=1" "lazy2=1,fuse_reduce=1" "lazy2=1,seg=128"; do
  SNARKVM_HIP_TUNING=$v timeout 200 python bench.py --workload proofs64 --no-cpu-baseline > "$O/p64_$v.json" 2> "$O/p64_$v.err"
  python - "$O/p64_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["concurrent_callers"]
    print(f"{sys.argv[2]:24s} lockstep {d['value']:.1f}/s ({d['ms_per_step']:.2f} ms) {({k: round(v, 2) for k, v in d['rank0_call_time_ms_per_proof'].items()})} g1 {d.get('g1_pairs_per_s_inside_msm_calls', 0):.3e} g2 {d.get('g2_pairs_per_s_inside_msm_calls', 0):.3e} | callers {c['value']:.1f}/s")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
