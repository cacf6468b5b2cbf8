#!/bin/bash
# Round 4, GPU session 1 (run from the repository root on the GPU box: gpurun -- 'bash tools/r04_session1.sh'):
# parity tests of the new paths, the scalar-read kernel variants, proofs64 in its two modes.  Everything lands in gpurun_out/r04_s1.
O=gpurun_out/r04_s1; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_proofs.py -x -q > $O/pytest_proofs.log 2>&1; echo "pytest_proofs rc=$?"; tail -3 $O/pytest_proofs.log
timeout 900 python -m pytest tests/test_gpu_multidevice.py -x -q -k "ab_switches and (hist or ntt_batch or coalesce or fuse_max_k)" > $O/pytest_ab.log 2>&1; echo "pytest_ab rc=$?"; tail -2 $O/pytest_ab.log
for v in "hist=2" "hist=1" "hist=3" "hist=2,hist_tiles=4" "hist=2,hist_tiles=8" "hist=2,hist_tiles=32" "hist=1"; do
  SNARKVM_HIP_TUNING=$v timeout 300 python bench.py --steps 4 --warmup 1 --no-extra-legs --no-cpu-baseline --ntt-steps 2 > "$O/bench_$v.json" 2> "$O/bench_$v.err"
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline_scalar_read"]
    print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:.2f}  scalar_read {d['phase_ms'].get('msm_scalar_read')} ms frac {r['frac']:.4f}  whole_phase {r['whole_phase']['ms']:.3f} ms  l1 {d['phase_ms'].get('msm_sort_level1')}  acc {d['phase_ms'].get('msm_accumulate')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
timeout 900 python bench.py --workload proofs64 > $O/proofs64.json 2> $O/proofs64.err; echo "proofs64 rc=$?"
SNARKVM_HIP_TUNING=coalesce=0 timeout 600 python bench.py --workload proofs64 --no-cpu-baseline > $O/proofs64_nocoalesce.json 2> $O/proofs64_nocoalesce.err
timeout 600 python bench.py --workload proofs64 --no-cpu-baseline --proof-group 64 --proof-workers 16 > $O/proofs64_g64_w16.json 2> $O/proofs64_g64_w16.err
timeout 600 python bench.py --workload proofs64 --no-cpu-baseline --proof-group 16 --proof-workers 4 > $O/proofs64_g16_w4.json 2> $O/proofs64_g16_w4.err
for f in proofs64 proofs64_nocoalesce proofs64_g64_w16 proofs64_g16_w4; do
  python - $O/$f.json $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["concurrent_callers"]
    print(f"{sys.argv[2]:22s} lockstep {d['value']:.1f} proofs/s ({d['ms_per_step']:.2f} ms, g1 {d['g1_pairs_per_s']:.3e}) times {({k: round(v, 2) for k, v in d['rank0_call_time_ms_per_proof'].items()})} | callers {c['value']:.1f} ({c['ms_per_proof']:.2f} ms) {({k: round(v, 2) for k, v in c['rank0_call_time_ms_per_proof'].items()})}")
    print("   checks:", d["checks"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
    print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
