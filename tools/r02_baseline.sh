#!/bin/bash
# round-2 baseline: GPU tests, C3 bench, wave stamps, C5 (10000 x 150 on one GPU) bench + kernel stats
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo pytest_exit=$? >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 200 python bench.py --steps 100 --warmup 10 > $O/bench_c3.log 2>&1 && tail -1 $O/bench_c3.log | cut -c1-1500
timeout -k 10 200 python tools/stamps.py > $O/stamps.log 2>&1; tail -30 $O/stamps.log
timeout -k 10 300 python bench.py --workload C5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5.log 2>&1 && tail -1 $O/bench_c5.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o prof -- python3 bench.py --workload C5 --steps 10 --warmup 2 --no-cpu-baseline --timed-only > $O/prof_c5.log 2>&1
cut -c1-200 $O/prof_c5/prof_kernel_stats.csv
