#!/bin/bash
# C5 (and C3) probe of library variants on one box: usage r02_ab_c5.sh <out-name> variant...
O=gpurun_out/$1; shift; rm -f $O
for v in "$@"; do
  CCAL_LIB=$PWD/build_variants/$v.so timeout -k 10 200 python tools/probe.py --workload C5 --steps 40 --tag $v 2>&1 | grep probe >> $O || exit 1
  CCAL_LIB=$PWD/build_variants/$v.so timeout -k 10 120 python tools/probe.py --tag $v 2>&1 | grep probe >> $O || exit 1
done
cat $O
