#!/bin/bash
# A/B of the shipped library against build_variants/<name>.so on one box, alternating: C3, C5, the 1250 x 150 share, C1.   tools/ab2.sh <out> variant...
O=gpurun_out/$1; shift; mkdir -p $(dirname $O); rm -f $O
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = default ]; then LIB=; else LIB=$PWD/build_variants/$v.so; fi
  CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload C3 --steps 300 --tag $v 2>&1 | grep probe >> $O || exit 1
  CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload C1 --steps 300 --tag $v 2>&1 | grep probe >> $O || exit 1
  [ $rep = 1 ] && { CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload C5 --steps 40 --tag $v 2>&1 | grep probe >> $O || exit 1; CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload C5 --images 1250 --steps 100 --tag $v 2>&1 | grep probe >> $O || exit 1; }
done; done; cat $O
