#!/bin/bash
# Round 6 evidence run (gpurun -- 'bash tools/r06_final.sh'): the DRIVER's bench command, the GPU suite, the proof-shaped workloads (transcript order as the headline,
# the same call list through the reference's own three symbols), the G2 tail sweep, the group NTT timing, rocprofv3 statistics and PMC passes, the soak.
# Every step runs under its own `timeout`: a hung step costs its limit, not the lease.
O=gpurun_out/r06_final; mkdir -p $O
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_command.json 2> $O/bench_driver_command.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_final/r06_bench_driver_command.json"))
    print("value", f"{d['value']:.4e}", "ms/step", round(d["ms_per_step"], 2), "witness-like", f"{d.get('value_witness_like', 0):.4e}", "scalar_read frac", round(d["roofline_scalar_read"]["frac"], 4),
          "ntt", f"{d['ntt_value']:.3e}", "mad_frac", round(d["alu_roofline"]["mad_frac"], 3), "cpu", f"{d['cpu_baseline']['value']:.3e}", d["cpu_baseline"].get("build"))
    for k in ("proof1", "proofs64", "concurrent_callers"):
        print(k, round(d[k]["value"], 1), d[k]["unit"], {kk: (round(vv.get("ms_per_proof", 0), 3) if isinstance(vv, dict) and "ms_per_proof" in vv else None) for kk, vv in d[k].items() if isinstance(vv, dict) and "ms_per_proof" in vv})
    f = d["proof1_ffi"]
    print("proof1_ffi stateless", round(f["stateless"]["ms_inside_the_three_symbols_per_proof"], 2), "cached", round(f["base_cache_16"]["ms_inside_the_three_symbols_per_proof"], 2), "resident", round(f["resident_ms_per_proof"], 2), f["checks"])
    print("checks", list(d["checks"].keys()))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r06_final/bench_driver_command.err").read()[-1500:])
PY
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 --timeout-method=thread > $O/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_pytest_gpu.log
timeout 400 python bench.py --workload proof1 > $O/r06_proof1.json 2> $O/proof1.err; echo "proof1 rc=$?"
timeout 400 python bench.py --workload proof1 --proof-mem torch --no-cpu-baseline > $O/r06_proof1_torch_buffers.json 2> $O/proof1_torch.err; echo "proof1 torch rc=$?"
timeout 400 python bench.py --workload proof1 --scalars witness --no-cpu-baseline > $O/r06_proof1_witness.json 2> $O/proof1_witness.err; echo "proof1 witness rc=$?"
timeout 600 python bench.py --workload proof1 --ffi-only > $O/r06_proof1_ffi_only.json 2> $O/proof1_ffi.err; echo "proof1 ffi-only rc=$?"
timeout 500 python bench.py --workload proofs64 > $O/r06_proofs64.json 2> $O/proofs64.err; echo "proofs64 rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_final/r06_proof1.json"))
    print("proof1 (transcript order, in-stream)", round(d["ms_per_step"], 3), "ms/proof |", {k: round(d[k]["ms_per_proof"], 3) for k in ("awaited_on_further_streams", "all_rounds_enqueued_at_once", "synchronous_commitments")}, d["checks"])
    f = json.load(open("gpurun_out/r06_final/r06_proof1_ffi_only.json"))
    print("ffi-only: stateless", round(f["stateless"]["ms_inside_the_three_symbols_per_proof"], 2), f["stateless"]["ms_by_step"], "| cached", round(f["base_cache_16"]["ms_inside_the_three_symbols_per_proof"], 2),
          "| resident", round(f["resident"]["ms_per_proof"], 2), "without g2", round(f["resident_without_g2"]["ms_per_proof"], 2), f["ratios"], f["checks"])
    d = json.load(open("gpurun_out/r06_final/r06_proofs64.json"))
    print("proofs64 lockstep", round(d["value"], 1), "proofs/s", d["checks"])
except Exception as e:
    print("proof parse failed", e)
PY
# the single proof without its G2 MSM in the four commitment modes
python - > $O/r06_proof1_without_g2.txt 2> $O/proof1_nog2.err <<'PY'
import sys, time
sys.path.insert(0, ".")
from snarkvm_amd import _lib, proofs
_lib.check(_lib.lib().snarkvm_hip_set_device(0))
keys = proofs.ProverKeys(proofs.ProofShape(lg_g2=0), tables=17, window_bits=15, mem="hip")
for label, mode, aw, ins in (("commitments awaited round by round on the scope's own stream (the headline order)", True, True, True), ("awaited on further streams", True, True, False),
                             ("all rounds enqueued at once", True, False, False), ("synchronous commitments", False, False, False)):
    ws = proofs.SingleProofWorkspace(keys)
    for s in range(4):
        proofs.replay_single(ws, s, None, mode, None, aw, ins)
    lat = []
    for s in range(32):
        t0 = time.perf_counter(); proofs.replay_single(ws, s, None, mode, None, aw, ins); lat.append(time.perf_counter() - t0)
    lat.sort()
    print(f"one proof at a time, NO G2 MSM (14 G1 results), {label}: mean {sum(lat) / len(lat) * 1e3:.3f} ms, median {lat[16] * 1e3:.3f}, min {lat[0] * 1e3:.3f}")
PY
cat $O/r06_proof1_without_g2.txt
# G2: the tail sweep (per-kernel times under rocprofv3) and the size table
TUNES="hex2=1,tail_quads=13 hex2=0,tail_quads=0 hex2=0,tail_quads=0,fold_threads2=128" timeout 900 bash tools/g2_tail.sh $O/g2tail > /dev/null 2>&1; grep -E "^==|msm_fold|msm_bitplane|ms_per_sync" $O/g2tail/summary.txt | cut -c1-200
timeout 200 python tools/bench_g2.py > $O/g2.md 2> $O/g2.err; cut -c1-75 $O/g2.md | tail -4
timeout 300 python tools/group_ntt_timing.py $O/r06_group_ntt_timing.md > /dev/null 2> $O/group_ntt.err; cat $O/r06_group_ntt_timing.md
# the soak: 5 GPU-minutes on one device, 2 minutes on two logical devices
timeout 700 python tools/soak.py --seconds 300 --threads 16 --out $O/r06_soak_300s.json > $O/soak300.log 2>&1; echo "soak rc=$?"; tail -c 900 $O/soak300.log
timeout 400 python tools/soak.py --seconds 120 --threads 16 --devices 2 --seed 7 --out $O/r06_soak_two_logical_devices.json > $O/soak2.log 2>&1; echo "soak 2 devices rc=$?"; tail -c 600 $O/soak2.log
# rocprofv3: kernel statistics + PMC passes of the headline (tools/profile_round.sh), trace of the proof
timeout 1500 bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; echo "profile rc=$?"
cp gpurun_out/r06prof/r06_* $O/ 2>/dev/null
F=$(find gpurun_out/r06prof/fetch -name "*.db" | head -1); W=$(find gpurun_out/r06prof/write -name "*.db" | head -1)
[ -n "$F" ] && [ -n "$W" ] && timeout 200 python tools/pmc_traffic.py $F $W "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh r06 (bench.py --steps 1 --no-pipeline, 2^24)" > $O/r06_pmc_traffic.json 2> $O/pmc_traffic.err
find gpurun_out/r06prof -name "*.db" -delete 2>/dev/null
ls $O
bash tools/logical_devices.sh r06 > $O/logical_devices.log 2>&1; tail -8 $O/logical_devices.log; cp gpurun_out/r06_logical/*.log $O/ 2>/dev/null
