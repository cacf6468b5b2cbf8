#!/bin/bash
# Round 4, GPU session 9: XCD-contiguous tile walk of the scatter kernels (A/B + parity), G2 fold workgroup size, dispatcher slots.
O=gpurun_out/r04_s9; mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multidevice.py -x -q --timeout 300 -k "not 2_25 and not two_rank" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in "xcd=1" "xcd=0"; do
  SNARKVM_HIP_TUNING=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --ntt-steps 2 > "$O/bench_$v.json" 2> "$O/bench_$v.err"
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    ph = d.get("phases_ms", {}) or d.get("phase_ms", {})
    rs = d.get("roofline_scalar_read", {})
    print(f"{sys.argv[2]:10s} ms/step {d['ms_per_step']:.2f} value {d['value']:.4e} scalar_read {rs.get('ms')} whole {rs.get('whole_phase', {}).get('ms') if isinstance(rs.get('whole_phase'), dict) else rs.get('whole_phase_ms')}", {k: round(v, 3) for k, v in ph.items() if 'sort' in k or 'acc' in k or 'digit' in k})
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
for v in "fold_threads2=128" "fold_threads2=256"; do echo "== $v"; SNARKVM_HIP_TUNING=$v timeout 120 python tools/bench_g2.py 2> "$O/g2_$v.err" | tee "$O/g2_$v.md" | cut -c1-400 | tail -3; done
g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_callers.cpp -o /tmp/bench_callers -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib || exit 1
for v in "coalesce_slots=1" "coalesce_slots=2" "coalesce_slots=3"; do echo "== $v"; GPU_MAX_HW_QUEUES=8 SNARKVM_HIP_TUNING=$v timeout 120 /tmp/bench_callers 4 8 16 32 2> "$O/callers_$v.err" | tee "$O/callers_$v.md"; done
for v in "xcd=1" "xcd=0"; do
  SNARKVM_HIP_TUNING=$v timeout 200 python bench.py --workload proofs64 --no-cpu-baseline > "$O/p64_$v.json" 2> "$O/p64_$v.err"
  python - "$O/p64_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["concurrent_callers"]
    print(f"{sys.argv[2]:24s} lockstep {d['value']:.1f}/s ({d['ms_per_step']:.2f} ms) {({k: round(v, 2) for k, v in d['rank0_call_time_ms_per_proof'].items()})} g1 {d.get('g1_pairs_per_s_inside_msm_calls', 0):.3e} | callers {c['value']:.1f}/s")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
