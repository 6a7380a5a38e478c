#!/bin/bash
O=gpurun_out/r02_fuzz_long; mkdir -p $O
for seed in ${SEEDS:-51 52 53 54 55 56 57 58}; do timeout -k 10 200 python tools/fuzz_parity.py --cases 500 --seed $seed > $O/fuzz_$seed.log 2>&1; echo "seed $seed exit $?"; grep "MISMATCH\|cases compared" $O/fuzz_$seed.log; done
