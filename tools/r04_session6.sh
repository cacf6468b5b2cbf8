#!/bin/bash
# Round 4, GPU session 6: G2 accumulation on the lazy Fq2 arithmetic (ffl2.hip.h): parity, A/B against lazy2=0, the proof replay.
O=gpurun_out/r04_s6; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu -k "g2 or G2 or proofs or lockstep or chunk_ring or serialize" > $O/pytest_g2.log 2>&1; echo "pytest_g2 rc=$?"; tail -3 $O/pytest_g2.log
for v in "lazy2=1" "lazy2=0"; do echo "== $v"; SNARKVM_HIP_TUNING=$v timeout 300 python tools/bench_g2.py 2> "$O/g2_$v.err" | tee "$O/g2_$v.md" | tail -6; done
for v in "lazy2=1" "lazy2=0"; do
  SNARKVM_HIP_TUNING=$v timeout 600 python bench.py --workload proofs64 --no-cpu-baseline > "$O/p64_$v.json" 2> "$O/p64_$v.err"
  python - "$O/p64_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["concurrent_callers"]
    print(f"{sys.argv[2]:10s} lockstep {d['value']:.1f}/s ({d['ms_per_step']:.2f} ms) {({k: round(v, 2) for k, v in d['rank0_call_time_ms_per_proof'].items()})} g2-in-calls {d.get('g2_pairs_per_s_inside_msm_calls', 0):.3e} | callers {c['value']:.1f}/s")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
