#!/bin/bash
# Round 5: the sweep with its wave-uniform operands in scalar registers (CCAL_SGPR_UNIFORMS) at two / three / four waves per SIMD
# (half-size tile for three and four), every arm on ONE box, alternating, two rounds.  Workloads: C3 (1000 x 54), C5 whole
# (10000 x 150), the per-GPU share of configs[4] at 8 GPUs (1250 x 150) with one and two waves per image, C1 (100 x 54).
#   tools/r05_occ_probe.sh <out-name> variant...      (variants: build_variants/<name>.so, "default" = the shipped library)
O=gpurun_out/$1; shift; mkdir -p gpurun_out; rm -f $O
run() {  # tag lib workload images steps [env...]
  local tag=$1 lib=$2 wl=$3 n=$4 steps=$5; shift 5
  env "$@" CCAL_LIB=$lib timeout -k 10 200 python tools/probe.py --workload $wl ${n:+--images $n} --steps $steps --tag "$tag" 2>&1 | grep probe >> $O || exit 1
}
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = default ]; then LIB=; else LIB=$PWD/build_variants/$v.so; fi
  run $v "$LIB" C3 "" 200 || exit 1
  run $v "$LIB" C5 "" 40 || exit 1
  run "$v/wpi1" "$LIB" C5 1250 100 CCAL_WPI=1 || exit 1
  run "$v/wpi2" "$LIB" C5 1250 100 CCAL_WPI=2 CCAL_WPI_CAP=1024 || exit 1
  run $v "$LIB" C1 "" 200 || exit 1
done; done
cat $O
