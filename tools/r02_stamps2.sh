#!/bin/bash
O=gpurun_out/r02_stamps; mkdir -p $O
CCAL_STAMP_BLOCK=7 timeout -k 10 200 python tools/stamps.py > $O/stamps_7.log 2>&1; tail -20 $O/stamps_7.log | head -17
timeout -k 10 120 python tools/probe.py 2>&1 | grep probe
