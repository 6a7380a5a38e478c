#!/bin/bash
O=gpurun_out/r02_probe6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trajectory or full_fit or c3_single or flag_words or golden" 2>&1 | tail -3
for w in C3 REF C1; do timeout -k 10 120 python tools/probe.py --workload $w ${IMG:+--images $IMG} 2>&1 | grep probe; done | tee $O/probe.log
timeout -k 10 120 python tools/probe.py --workload REF --images 1000 2>&1 | grep probe | tee -a $O/probe.log
