#!/bin/bash
O=gpurun_out/r02_tests; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 "$@" > $O/pytest.log 2>&1; echo exit=$? >> $O/pytest.log; tail -40 $O/pytest.log
