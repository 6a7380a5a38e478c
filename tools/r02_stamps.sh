#!/bin/bash
O=gpurun_out/r02_stamps; mkdir -p $O
for B in 7 0 249; do CCAL_STAMP_BLOCK=$B timeout -k 10 200 python tools/stamps.py > $O/stamps_$B.log 2>&1; echo "== block $B"; tail -20 $O/stamps_$B.log | head -17; done
timeout -k 10 120 python tools/probe.py 2>&1 | grep probe
