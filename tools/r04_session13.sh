#!/bin/bash
# Round 4, GPU session 13: the GPU suite on the final build under two logical devices; bench default again (shader-clock sampler fixed).
O=gpurun_out/r04_s13; mkdir -p $O
export TMPDIR=/tmp
timeout 420 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_s13/r04_bench_default.json"))
print("value", f"{d['value']:.4e}", "ms/step", round(d["ms_per_step"], 2), "scalar_read", round(d["roofline_scalar_read"]["frac"], 4), "whole", round(d["roofline_scalar_read"]["whole_phase"]["ms"], 3),
      "ntt", f"{d['ntt_value']:.3e}", "mad_frac", round(d["alu_roofline"]["mad_frac"], 3), d["alu_roofline"].get("sclk_during_timed_steps"), d["alu_roofline"].get("mad_frac_at_sustained_clock"),
      "ffi 2^24", round(d["end_to_end_ffi"]["snarkvm_msm_2p24"]["call_ms"], 1), "reg+host", round(d["end_to_end_ffi"]["snarkvm_msm_2p24"]["registered_bases_host_scalars_ms"], 1))
PY
SNARKVM_HIP_DEVICES=0,0 timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "not 2_24 and not 2_22 and not 2_25 and not bench_" > $O/r04_pytest_gpu_two_logical_devices.log 2>&1; echo "two-device suite rc=$?"; tail -3 $O/r04_pytest_gpu_two_logical_devices.log
