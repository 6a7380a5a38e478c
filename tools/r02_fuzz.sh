#!/bin/bash
O=gpurun_out/r02_fuzz; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo pytest_exit=$? >> $O/pytest.log; tail -4 $O/pytest.log
for seed in ${SEEDS:-21 22 23}; do timeout -k 10 280 python tools/fuzz_parity.py --cases 500 --seed $seed > $O/fuzz_$seed.log 2>&1; echo "seed $seed exit $?"; tail -3 $O/fuzz_$seed.log; done
