#!/bin/bash
O=gpurun_out/r02_bench; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bench_rehearsal.py tests/test_gpu_parity.py -m gpu -x -q -k "bench or maps or errors or calibration or fit_model" > $O/pytest.log 2>&1; echo exit=$? >> $O/pytest.log; tail -15 $O/pytest.log
timeout -k 10 300 python bench.py --steps 50 --warmup 5 > $O/bench1.log 2>&1; tail -1 $O/bench1.log | cut -c1-6000
