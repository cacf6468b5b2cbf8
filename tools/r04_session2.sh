#!/bin/bash
# Round 4, GPU session 2: the wide scalar-read kernel in the product, strided Fr passes in the lock-step replay, coalescer settings.
O=gpurun_out/r04_s2; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_proofs.py tests/test_gpu_poly.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in "hist=2" "hist=1"; do
  SNARKVM_HIP_TUNING=$v timeout 300 python bench.py --steps 4 --warmup 1 --no-extra-legs --no-cpu-baseline --ntt-steps 2 > "$O/bench_$v.json" 2> "$O/bench_$v.err"
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline_scalar_read"]
    print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:.2f}  scalar_read {d['phase_ms'].get('msm_scalar_read')} ms frac {r['frac']:.4f}  whole_phase {r['whole_phase']['ms']:.3f} ms  l1 {d['phase_ms'].get('msm_sort_level1')}  acc {d['phase_ms'].get('msm_accumulate')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
run_p64() {  # name, tuning, extra args
  SNARKVM_HIP_TUNING="$2" timeout 600 python bench.py --workload proofs64 --no-cpu-baseline $3 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["concurrent_callers"]
    print(f"{sys.argv[2]:26s} lockstep {d['value']:.1f}/s ({d['ms_per_step']:.2f} ms) g1-in-msm {d.get('g1_pairs_per_s_inside_msm_calls', 0):.3e} {({k: round(v, 2) for k, v in d['rank0_call_time_ms_per_proof'].items()})} | callers {c['value']:.1f}/s ({c['ms_per_proof']:.2f} ms) {({k: round(v, 2) for k, v in c['rank0_call_time_ms_per_proof'].items()})}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
run_p64 p64_default "" ""
run_p64 p64_g64 "" "--proof-group 64"
run_p64 p64_us0 "coalesce_us=0" ""
run_p64 p64_us150 "coalesce_us=150" ""
run_p64 p64_us400 "coalesce_us=400" ""
run_p64 p64_w12 "" "--proof-workers 12"
run_p64 p64_w16_us150 "coalesce_us=150" "--proof-workers 16"
run_p64 p64_k16 "fuse_max_k=16" ""
run_p64 p64_k128 "fuse_max_k=128" "--proof-group 64"
