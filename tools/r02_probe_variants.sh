#!/bin/bash
# usage: r02_probe_variants.sh <out-name> <images> variant...   (libraries from tools/variants.py)
O=gpurun_out/$1; N=$2; shift; shift; mkdir -p gpurun_out; rm -f $O
for v in "$@"; do
  CCAL_LIB=$PWD/build_variants/$v.so timeout -k 10 120 python tools/probe.py --images $N --tag $v 2>&1 | grep probe >> $O || exit 1
done
cat $O
