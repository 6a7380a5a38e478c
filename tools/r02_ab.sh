#!/bin/bash
# A/B of two library builds on one box: usage r02_ab.sh <out-name> variantA variantB   (C3 and C5, alternating)
O=gpurun_out/$1; shift; rm -f $O
for rep in 1 2; do for v in "$@"; do
  CCAL_LIB=$PWD/build_variants/$v.so timeout -k 10 120 python tools/probe.py --tag $v 2>&1 | grep probe >> $O || exit 1
  CCAL_LIB=$PWD/build_variants/$v.so timeout -k 10 200 python tools/probe.py --workload C5 --steps 40 --tag $v 2>&1 | grep probe >> $O || exit 1
done; done
cat $O
