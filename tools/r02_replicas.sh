#!/bin/bash
# launch time against the number of accumulator replicas (CCAL_ACC_REPLICAS), one box
O=gpurun_out/replicas.txt; rm -f $O
for rep in 1 2; do for r in 8 16 32; do
  CCAL_ACC_REPLICAS=$r timeout -k 10 120 python tools/probe.py --tag "R$r" 2>&1 | grep probe >> $O || exit 1
done; done
cat $O
