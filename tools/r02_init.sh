#!/bin/bash
# init kernels: parity tests that cover them + the bench's init block + rocprof kernel means
O=gpurun_out/r02_init; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "initialisation or init or example or fit_model or full_calibration or flag_words or edge" > $O/pytest.log 2>&1; echo pytest_exit=$?; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; echo "bench exit $?"
grep "k_init" $O/prof/prof_kernel_stats.csv | cut -c1-160
grep -o '"init": {[^}]*}' $O/bench.log | tail -1
