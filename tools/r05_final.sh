#!/bin/bash
# Round 5 evidence run (gpurun -- 'bash tools/r05_final.sh'): the DRIVER's bench command (--gpus 1 --steps 20 --warmup 5, not bench.py's defaults: round 4's
# evidence runs used the defaults and never saw what the driver saw), the GPU suite, the proof-shaped workloads, the single-proof timeline,
# rocprofv3 statistics and PMC passes, the side tables of profiles/r05_summary.md.
O=gpurun_out/r05_final; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_driver_command.json 2> $O/bench_driver_command.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_final/r05_bench_driver_command.json"))
    print("value", f"{d['value']:.4e}", "ms/step", round(d["ms_per_step"], 2), "scalar_read frac", round(d["roofline_scalar_read"]["frac"], 4), "ntt", f"{d['ntt_value']:.3e}",
          "mad_frac", round(d["alu_roofline"]["mad_frac"], 3), "tables1", round(d["tables1_ms_per_step"], 2), d["tables1_workspace_growth"], "cpu", f"{d['cpu_baseline']['value']:.3e}")
    print("ffi", {k: round(v.get("call_ms", v.get("ms", 0)), 2) for k, v in d["end_to_end_ffi"].items() if isinstance(v, dict)})
    for k in ("proof1", "proofs64", "concurrent_callers"):
        print(k, round(d[k]["value"], 1), d[k]["unit"], d[k]["checks"])
    print("checks", list(d["checks"].keys()))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r05_final/bench_driver_command.err").read()[-1500:])
PY
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05_pytest_gpu.log
timeout 300 python bench.py --workload proof1 > $O/r05_proof1.json 2> $O/proof1.err; echo "proof1 rc=$?"
timeout 400 python bench.py --workload proofs64 > $O/r05_proofs64.json 2> $O/proofs64.err; echo "proofs64 rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_final/r05_proof1.json"))
    print("proof1", round(d["ms_per_step"], 3), "ms/proof", d["latency"], "| awaited round by round:", round(d["commitments_awaited_round_by_round"]["ms_per_proof"], 3),
          "| awaited, in-stream:", round(d["commitments_awaited_in_stream"]["ms_per_proof"], 3),
          "| sync commitments:", round(d["other_commitment_mode"]["ms_per_proof"], 3), d["checks"])
    d = json.load(open("gpurun_out/r05_final/r05_proofs64.json"))
    c = d["concurrent_callers"]
    print("proofs64 lockstep", round(d["value"], 1), "proofs/s; callers", round(c["value"], 1), c["coalescer"], "| async scope", round(c["one_asynchronous_scope_per_proof"]["value"], 1),
          "| serial", round(c["one_synchronous_call_per_step"]["value"], 1), d["checks"])
except Exception as e:
    print("proof parse failed", e)
PY
# the single proof without its G2 MSM (what round 4's 10.8 ms figure was measured on) and the timeline of both commitment modes
python - > $O/r05_proof1_without_g2.txt 2> $O/proof1_nog2.err <<'PY'
import sys, time
sys.path.insert(0, ".")
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from snarkvm_amd import _lib, proofs
torch.cuda.set_device(0)
_lib.check(_lib.lib().snarkvm_hip_set_device(0))
keys = proofs.ProverKeys(proofs.ProofShape(lg_g2=0), tables=17, window_bits=15)
for label, mode, aw, ins in (("asynchronous commitments", True, False, False), ("commitments awaited round by round (snarkvm_hip_scope_collect), on further streams", True, True, False),
                             ("commitments awaited round by round, on the scope's own stream (SNARKVM_HIP_SCOPE_MSM_IN_STREAM)", True, True, True), ("synchronous commitments", False, False, False)):
    ws = proofs.SingleProofWorkspace(keys)
    for s in range(4):
        proofs.replay_single(ws, s, None, mode, None, aw, ins)
    lat = []
    for s in range(32):
        t0 = time.perf_counter(); proofs.replay_single(ws, s, None, mode, None, aw, ins); lat.append(time.perf_counter() - t0)
    lat.sort()
    print(f"one proof at a time, NO G2 MSM (14 G1 results), {label}: mean {sum(lat) / len(lat) * 1e3:.3f} ms, median {lat[16] * 1e3:.3f}, min {lat[0] * 1e3:.3f}")
PY
cat $O/r05_proof1_without_g2.txt
for mode in async sync; do
  extra=""; [ $mode = sync ] && extra="--sync-msm"
  timeout 300 rocprofv3 --kernel-trace -d $O/trace_$mode -o t -- python tools/proof1_timeline.py run $O/marks_$mode.json $extra --proofs 16 > $O/trace_$mode.log 2>&1; echo "trace $mode rc=$?"
  DB=$(find $O/trace_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/proof1_timeline.py report $DB $O/marks_$mode.json > $O/r05_proof1_timeline_$mode.md 2> $O/timeline_$mode.err
done
find $O -name "*.db" -delete
timeout 1500 bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; echo "profile rc=$?"
cp gpurun_out/r05prof/r05_* $O/ 2>/dev/null
# side tables
timeout 200 python tools/reg_host_scalars.py 16 20 22 24 > $O/reg_host_scalars.md 2> $O/reg_host_scalars.err; cat $O/reg_host_scalars.md
SNARKVM_HIP_BASE_CACHE=0 timeout 200 python tools/ffi_msm_sweep.py 16 20 22 24 > $O/ffi_msm.md 2> $O/ffi_msm.err; cat $O/ffi_msm.md
timeout 100 python tools/phase_ffi.py > $O/phase_ffi.md 2> $O/phase_ffi.err; cat $O/phase_ffi.md | head -40
timeout 150 python tools/bench_g2.py > $O/g2.md 2> $O/g2.err; cut -c1-75 $O/g2.md | tail -4
SNARKVM_HIP_TUNING=pair2=0 timeout 150 python tools/bench_g2.py > $O/g2_pair0.md 2> $O/g2_pair0.err; cut -c1-75 $O/g2_pair0.md | tail -4
timeout 120 tools/exp/mfma_reduction > $O/r05_mfma_reduction.txt 2>&1; cat $O/r05_mfma_reduction.txt
timeout 200 python tools/tables1_cliff.py > $O/r05_tables1_cliff.md 2> $O/tables1_cliff.err; cat $O/r05_tables1_cliff.md
g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_callers.cpp -o /tmp/bench_callers -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib && GPU_MAX_HW_QUEUES=8 timeout 120 /tmp/bench_callers 1 8 32 > $O/callers.md 2> $O/callers.err; cat $O/callers.md
python - <<'PY'
import sys
sys.path.insert(0, ".")
from snarkvm_amd import synthetic
open("/tmp/g2pts.bin", "wb").write(synthetic.g2_points(1 << 16).tobytes())
PY
g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_proof_callers.cpp -o /tmp/bench_proof_callers -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib && {
  for mode in "" "--scope" "--scope-await" "--scope-await-in-stream" "--scope-sync"; do
    GPU_MAX_HW_QUEUES=8 timeout 200 /tmp/bench_proof_callers - $mode 1 8 16 > "$O/proof_callers_nog2$mode.md" 2> "$O/proof_callers_nog2$mode.err"; cat "$O/proof_callers_nog2$mode.md"
  done
  GPU_MAX_HW_QUEUES=8 timeout 200 /tmp/bench_proof_callers /tmp/g2pts.bin --scope-sync 1 8 16 > "$O/proof_callers_g2--scope-sync.md" 2>&1; cat "$O/proof_callers_g2--scope-sync.md"
}
ls $O
bash tools/logical_devices.sh r05 > $O/logical_devices.log 2>&1; tail -8 $O/logical_devices.log; cp gpurun_out/r05_logical/*.log $O/ 2>/dev/null
