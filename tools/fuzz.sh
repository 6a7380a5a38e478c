#!/bin/bash
# random problems, device vs oracle (tools/fuzz_parity.py), 500 cases per seed.   SEEDS="61 62" tools/fuzz.sh <tag>
O=gpurun_out/${1:-fuzz}; mkdir -p $O
for seed in ${SEEDS:-61 62 63}; do timeout -k 10 280 python tools/fuzz_parity.py --cases 500 --seed $seed > $O/fuzz_$seed.log 2>&1; echo "seed $seed exit $?"; grep "MISMATCH\|cases compared" $O/fuzz_$seed.log; done
