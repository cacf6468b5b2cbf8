#!/bin/bash
# sweep of the G2 tail variants on the GPU box: bash tools/g2_tail.sh <out dir> ; per-kernel times from rocprofv3 --kernel-trace --stats
out=${1:-gpurun_out/g2_tail}; mkdir -p $out; export TMPDIR=/tmp
TUNES=${TUNES:-"hex2=1,tail_quads=13 hex2=0,tail_quads=13 hex2=1,tail_quads=0 hex2=0,tail_quads=0 hex2=1,tail_quads=15"}
for tune in $TUNES; do
  tag=$(echo $tune | tr ',=' '__')
  SNARKVM_HIP_TUNING=$tune timeout 120 python tools/g2_tail.py > $out/$tag.json 2> $out/$tag.err
  (cd /tmp && SNARKVM_HIP_TUNING=$tune timeout 180 rocprofv3 --kernel-trace --stats -d /tmp/g2prof_$tag -o p -- python $OLDPWD/tools/g2_tail.py > /dev/null 2> $OLDPWD/$out/$tag.prof.err)
  db=$(find /tmp/g2prof_$tag -name "*.db" | head -1)
  csv=$(find /tmp/g2prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tune" >> $out/summary.txt
  cat $out/$tag.json >> $out/summary.txt
  if [ -n "$csv" ]; then grep -E "msm_fold_kernel|msm_bitplane_kernel|msm_accumulate_pair2|Name" $csv | cut -c1-220 >> $out/summary.txt; fi
  if [ -n "$db" ]; then python tools/rocprof_summary.py stats $db 2>/dev/null | grep -E "msm_fold|msm_bitplane|accumulate_pair2|kernel " | cut -c1-200 >> $out/summary.txt; fi
done
cat $out/summary.txt
