#!/bin/bash
O=gpurun_out/r02_probe4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trajectory or full_fit or c3_single or gram" 2>&1 | tail -3
timeout -k 10 120 python tools/probe.py 2>&1 | grep probe | tee $O/probe.log
timeout -k 10 120 python tools/probe.py --workload C5 --steps 30 2>&1 | grep probe | tee -a $O/probe.log
timeout -k 10 120 python tools/probe.py --workload C5 --images 1250 --steps 50 2>&1 | grep probe | tee -a $O/probe.log
