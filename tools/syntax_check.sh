#!/bin/bash
# host + device syntax check of every translation unit (seconds; the full build takes minutes): bash tools/syntax_check.sh
cd "$(dirname "$0")/../snarkvm_amd/csrc" || exit 1
for f in api.hip api_fr.hip api_g2.hip api_serde.hip tail_g1.hip tail_g2.hip tail_g2_planes.hip tail_g2_fix.hip; do
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fsyntax-only -Wno-unused-result -Wno-pass-failed $f 2>&1 | grep -E "error" -A4 | head -40 | sed "s|^|$f: |") &
done
wait
