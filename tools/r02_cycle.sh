#!/bin/bash
# round-2 GPU cycle: parity tests, C3 bench, wave stamps, C5 bench.  usage: r02_cycle.sh <tag> [pytest args]
TAG=${1:-x}; shift
O=gpurun_out/r02_$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q "$@" > $O/pytest_gpu.log 2>&1; echo pytest_exit=$? >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
timeout -k 10 200 python bench.py --steps 100 --warmup 10 > $O/bench_c3.log 2>&1; tail -1 $O/bench_c3.log | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('C3 value %.4g  ms/step %.4f' % (d['value'], d['ms_per_step']), d['lm'], d['init'])
    print({k: d['roofline'][k] for k in ('avg_launch_us', 'avg_launch_us_back_to_back', 'frac')})
except Exception as e:
    print('bench parse failed', e)
"
timeout -k 10 200 python tools/stamps.py > $O/stamps.log 2>&1; tail -22 $O/stamps.log | head -16
timeout -k 10 300 python bench.py --workload C5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('C5 value %.4g  ms/step %.4f' % (d['value'], d['ms_per_step']), d['lm'])
    print({k: d['roofline'][k] for k in ('avg_launch_us', 'avg_launch_us_back_to_back', 'frac')})
except Exception as e:
    print('bench parse failed', e)
"
