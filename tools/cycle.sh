#!/bin/bash
# One GPU cycle (run through gpurun from the repository root): parity tests, then the C3 and C5 bench lines.
#   tools/cycle.sh <tag> [pytest args]        output under gpurun_out/<tag>/
TAG=${1:-cycle}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=10 "$@" > $O/pytest_gpu.log 2>&1; echo pytest_exit=$? >> $O/pytest_gpu.log; tail -18 $O/pytest_gpu.log
grep -q "pytest_exit=0" $O/pytest_gpu.log || exit 1
show() { tail -1 $1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('$2 value %.4g  ms/step %.4f' % (d['value'], d['ms_per_step']), d['lm'], d.get('init'))
    print({k: d['roofline'].get(k) for k in ('avg_launch_us', 'avg_launch_us_back_to_back', 'frac', 'fp64_frac', 'latency_floor_us', 'frac_of_floor')})
    print({k: d.get(k) for k in ('host_overhead_us_per_step', 'eager_ms_per_step')})
except Exception as e:
    print('bench parse failed', e)
"; }
timeout -k 10 300 python bench.py --steps 100 --warmup 10 > $O/bench_c3.log 2> $O/bench_c3.err && show $O/bench_c3.log C3
timeout -k 10 300 python bench.py --workload C5 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c5.log 2> $O/bench_c5.err && show $O/bench_c5.log C5
