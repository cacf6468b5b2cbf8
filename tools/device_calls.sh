#!/bin/bash
# Which kernels of the built library still call a device function (s_swappc)?  bash tools/device_calls.sh [object dir]  (default: the newest build's objects)
# Round 6: calls inside the 512-register Fq2 tail kernels came back with caller registers overwritten - the tail units must list NOTHING here.
D=${1:-$(ls -td /tmp/snarkvm_hip_$(id -u)/obj/*/ | head -1)}
B=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
for o in $D/*.o; do
  t=$(basename $o .o)
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin $o $T/fat.bin && $B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$t.co 2>/dev/null
  echo "== $t"
  $B/llvm-objdump -d $T/$t.co | awk '/^[0-9a-f]+ <.*>:$/ {name=$2} /s_swappc/ {c[name]++} END {for (k in c) print c[k], k}' | sort -rn | sed 's/[<>:]//g' | while read n k; do echo "  $n call(s)  $(echo $k | c++filt | cut -c1-150)"; done
done
rm -rf $T
