#!/bin/bash
O=gpurun_out/r02_probe7; mkdir -p $O
for v in default noilp ilp_unroll memsched; do
  if [ $v = default ]; then unset CCAL_LIB; else export CCAL_LIB=$PWD/build_variants/$v.so; fi
  [ $v != default ] && [ ! -f "$CCAL_LIB" ] && continue
  timeout -k 10 120 python tools/probe.py --tag $v 2>&1 | grep probe
  timeout -k 10 120 python tools/probe.py --tag $v --workload C5 --steps 30 2>&1 | grep probe
done | tee $O/probe.log
