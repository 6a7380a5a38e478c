#!/bin/bash
for v in default minw1 default minw1; do
  if [ $v = default ]; then unset CCAL_LIB; else export CCAL_LIB=$PWD/build_variants/$v.so; fi
  timeout -k 10 120 python tools/probe.py --tag $v 2>&1 | grep probe
done
