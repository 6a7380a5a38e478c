#!/bin/bash
O=gpurun_out/r02_stamps; mkdir -p $O
CCAL_STAMPS_MEM=1 CCAL_STAMP_BLOCK=7 timeout -k 10 200 python tools/stamps.py > $O/stamps_mem7.log 2>&1; grep -B1 -A3 "entry ->" $O/stamps_mem7.log | tail -12
CCAL_STAMPS_MEM=1 CCAL_STAMP_BLOCK=100 timeout -k 10 200 python tools/stamps.py > $O/stamps_mem100.log 2>&1; grep -A2 "entry ->" $O/stamps_mem100.log | tail -6
