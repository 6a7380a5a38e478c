#!/bin/bash
O=gpurun_out/r02_probe; mkdir -p $O
for R in 1 2 4 8 16 32; do CCAL_ACC_REPLICAS=$R timeout -k 10 120 python tools/probe.py 2>&1 | grep probe; done | tee $O/probe.log
CCAL_LIB=$PWD/build_variants/store.so timeout -k 10 120 python tools/probe.py --tag store-not-atomic 2>&1 | grep probe | tee -a $O/probe.log
timeout -k 10 200 python tools/stamps.py > $O/stamps.log 2>&1; tail -22 $O/stamps.log | head -16
