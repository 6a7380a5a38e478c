#!/bin/bash
O=gpurun_out/r02_stamps; mkdir -p $O
for R in 2 8 32; do echo "== R=$R"; CCAL_ACC_REPLICAS=$R CCAL_STAMP_BLOCK=7 timeout -k 10 200 python tools/stamps.py > $O/stamps_R$R.log 2>&1; grep "prologue:" $O/stamps_R$R.log | tail -1; done
