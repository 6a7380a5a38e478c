#!/bin/bash
# A/B of library variants (tools/variants.py -> build_variants/<name>.so; "default" = the shipped library) on ONE box --
# boxes differ by a few per cent -- alternating, two rounds.   tools/ab.sh <out-name> "<workloads, e.g. C3 C5>" variant...
# Environment of the caller is passed on (e.g. CCAL_ACC_REPLICAS, CCAL_NO_GRAPH).
O=gpurun_out/$1; WLS=$2; shift; shift; mkdir -p gpurun_out; rm -f $O
for rep in 1 2; do for v in "$@"; do for wl in $WLS; do
  if [ "$v" = default ]; then LIB=; else LIB=$PWD/build_variants/$v.so; fi
  steps=200; [ $wl = C5 ] && steps=40
  CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload $wl --steps $steps --tag $v 2>&1 | grep probe >> $O || exit 1
done; done; done
cat $O
