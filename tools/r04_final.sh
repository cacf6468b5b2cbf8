#!/bin/bash
# Round 4 evidence run (gpurun -- 'bash tools/r04_final.sh'): the driver's bench command, the GPU suite, proofs64, rocprofv3 statistics and PMC passes.
O=gpurun_out/r04_final; mkdir -p $O
export TMPDIR=/tmp
timeout 420 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_final/r04_bench_default.json"))
    print("value", f"{d['value']:.4e}", "ms/step", round(d["ms_per_step"], 2), "scalar_read frac", round(d["roofline_scalar_read"]["frac"], 4), "ntt", f"{d['ntt_value']:.3e}",
          "mad_frac", round(d["alu_roofline"]["mad_frac"], 3), "ffi 2^24", round(d["end_to_end_ffi"]["snarkvm_msm_2p24"]["call_ms"], 1), "cpu", f"{d['cpu_baseline']['value']:.3e}")
    print("checks", list(d["checks"].keys()))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r04_final/bench_default.err").read()[-1500:])
PY
timeout 1100 python -m pytest tests -m gpu -q -x --timeout 300 > $O/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04_pytest_gpu.log
timeout 300 python bench.py --workload proofs64 > $O/r04_proofs64.json 2> $O/proofs64.err; echo "proofs64 rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_final/r04_proofs64.json"))
    c = d["concurrent_callers"]
    print("lockstep", round(d["value"], 1), "proofs/s", round(d["ms_per_step"], 2), "ms; g1 in calls", f"{d['g1_pairs_per_s_inside_msm_calls']:.3e}", "callers", round(c["value"], 1), c["coalescer"])
    print(d["checks"])
except Exception as e:
    print("proofs64 parse failed", e)
PY
timeout 1500 bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; echo "profile rc=$?"
cp gpurun_out/r04prof/r04_* $O/ 2>/dev/null; ls $O
# side tables of the round (profiles/r04_summary.md)
timeout 200 python tools/reg_host_scalars.py 16 20 22 24 > $O/reg_host_scalars.md 2> $O/reg_host_scalars.err; cat $O/reg_host_scalars.md
SNARKVM_HIP_BASE_CACHE=0 timeout 200 python tools/ffi_msm_sweep.py 16 20 22 24 > $O/ffi_msm.md 2> $O/ffi_msm.err; cat $O/ffi_msm.md
timeout 150 python tools/bench_g2.py > $O/g2.md 2> $O/g2.err; cut -c1-60 $O/g2.md | tail -4
g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_callers.cpp -o /tmp/bench_callers -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib && GPU_MAX_HW_QUEUES=8 timeout 120 /tmp/bench_callers 1 8 32 > $O/callers.md 2> $O/callers.err; cat $O/callers.md
