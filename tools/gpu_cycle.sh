#!/bin/bash
# one GPU cycle: parity tests, bench, rocprof kernel stats (run through gpurun from the repo root)
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo pytest_exit=$? >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 200 python bench.py "$@" > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('value %.4g  ms/step %.4f' % (d['value'], d['ms_per_step']), d['lm'], d['init'])
    print(d['roofline']); print(d.get('cpu_baseline'))
except Exception as e:
    print('bench parse failed', e)
"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
cut -c1-160 gpurun_out/prof/prof_kernel_stats.csv
