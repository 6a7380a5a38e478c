#!/bin/bash
O=gpurun_out/r02_probe2; mkdir -p $O
for v in 0 1; do HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python tools/probe.py --tag kernarg_dev=$v 2>&1 | grep probe; done | tee $O/probe.log
for R in 8 16; do CCAL_ACC_REPLICAS=$R timeout -k 10 120 python tools/probe.py --tag R$R 2>&1 | grep probe; done | tee -a $O/probe.log
CCAL_NO_GRAPH=1 timeout -k 10 120 python tools/probe.py --tag nograph 2>&1 | grep probe | tee -a $O/probe.log
