#!/bin/bash
# Round-3 side evidence on the GPU box (outputs under gpurun_out/r03b/): arithmetic ceilings, small-NTT tile rule A/B, G2 phases,
# NTT / G2 parity subset, multi-device split timelines, proofs64.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r03b
mkdir -p $O
timeout 300 tools/ecbench > $O/ecbench.txt 2>&1
for k in 1 256 512; do SNARKVM_HIP_TUNING=ntt_min_tiles=$k timeout 120 python tools/ntt_small.py > $O/ntt_small_$k.md 2>&1; done
timeout 300 python tools/bench_g2.py > $O/g2.md 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ntt or g2 or polymul" > $O/pytest_ntt_g2.log 2>&1
timeout 300 python -m pytest tests/test_gpu_poly.py tests/test_gpu_sonic.py -q -x > $O/pytest_poly_sonic.log 2>&1
timeout 900 python tools/multidevice_split.py 24 > $O/multidevice_split.md 2>&1
timeout 300 python bench.py --workload proofs64 --proof-workers 8 > $O/proofs64.json 2> $O/proofs64.err
SNARKVM_HIP_TUNING=ntt_min_tiles=1 timeout 300 python bench.py --workload proofs64 --proof-workers 8 > $O/proofs64_oldtiles.json 2> $O/proofs64_oldtiles.err
tail -2 $O/pytest_ntt_g2.log $O/pytest_poly_sonic.log; tail -3 $O/multidevice_split.md; cat $O/ntt_small_*.md; grep -v amdgpu $O/g2.md | cut -c1-900; tail -22 $O/ecbench.txt | cut -c1-200
