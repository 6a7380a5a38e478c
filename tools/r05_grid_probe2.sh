#!/bin/bash
# Round 5: the NON-looping sweep kernel (126-136 VGPRs with its uniform operands in scalar registers, no scratch) on an
# oversubscribed grid (one workgroup per image group, CCAL_MAX_GRID=4096: 2500 workgroups for 10000 images, dispatched as slots
# free up) at two / three / four workgroups per CU, against the looping kernel on a resident grid.
O=gpurun_out/r05c/grid2.txt; mkdir -p gpurun_out/r05c; rm -f $O
for rep in 1 2; do
for cfg in "default:512" "default:4096" "sg:4096" "sg3:4096" "sg4:4096"; do
  v=${cfg%%:*}; g=${cfg##*:}
  if [ "$v" = default ]; then LIB=; else LIB=$PWD/build_variants/$v.so; fi
  CCAL_MAX_GRID=$g CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload C5 --steps 40 --tag "$v/grid$g" 2>&1 | grep probe >> $O || exit 1
done; done; cat $O
