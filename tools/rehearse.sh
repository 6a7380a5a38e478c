#!/bin/bash
# The driver's N-rank command -- `python3 bench.py --gpus N`, no launcher -- rehearsed with the ranks sharing the one GPU of
# the box (not a measurement).       tools/rehearse.sh <N> [bench args]
N=${1:-4}; shift
O=gpurun_out/rehearse; mkdir -p $O
CCAL_P2P_TIMEOUT_MS=8000 timeout -k 10 900 python3 bench.py --gpus $N --steps 10 --warmup 2 --share-gpu "$@" > $O/n$N.out 2> $O/n$N.err
echo exit=$?; grep -v "^W1004\|^\[W\|Gloo" $O/n$N.err | tail -15 | cut -c1-400; tail -1 $O/n$N.out | cut -c1-3000
