export ROOT=/root/repo
for lib in "$@"; do
for t in "" "hex2=0,tail_quads=0" "fold_small2=128"; do
  echo "lib=$lib"; SNARKVM_HIP_LIB=$lib DISTINCT=16 SNARKVM_HIP_TUNING="$t" timeout 120 python tools/exp/repeat_msm.py g2 10 40 17 15 2>&1 | tail -1
done
SNARKVM_HIP_LIB=$lib timeout 120 python tools/exp/repeat_msm.py g2 10 300 17 15 2>&1 | tail -1
SNARKVM_HIP_LIB=$lib DISTINCT=16 timeout 120 python tools/exp/repeat_msm.py g2 10 40 22 12 2>&1 | tail -1
done
