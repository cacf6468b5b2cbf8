#!/bin/bash
# Round 4, GPU session 5: snarkvm_msm over host buffers - lanes of the chunk ring x chunk size; coalescer statistics of the 8-caller proof replay.
O=gpurun_out/r04_s5; mkdir -p $O
export TMPDIR=/tmp
for ring in 3 4 6; do for lg in 19 20 21; do
  v="ring_lanes=$ring,msm_chunk_lg=$lg"
  SNARKVM_HIP_TUNING=$v SNARKVM_HIP_BASE_CACHE=0 timeout 300 python tools/bench_ffi.py 20 22 24 2> "$O/ffi_$v.err" > "$O/ffi_$v.md"
  echo "$v: $(grep -E '^\| (20|22|24) ' "$O/ffi_$v.md" | awk -F'|' '{printf "2^%s %s ms; ", $2, $3}')"
done; done
timeout 600 python bench.py --workload proofs64 --no-cpu-baseline > $O/p64.json 2> $O/p64.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_s5/p64.json"))
c = d["concurrent_callers"]
print("lockstep", round(d["value"], 1), "callers", round(c["value"], 1), c.get("coalescer"))
PY
timeout 600 python bench.py --workload proofs64 --no-cpu-baseline --proof-workers 16 > $O/p64_w16.json 2> $O/p64_w16.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_s5/p64_w16.json"))
c = d["concurrent_callers"]
print("w16: lockstep", round(d["value"], 1), "callers", round(c["value"], 1), c.get("coalescer"))
PY
