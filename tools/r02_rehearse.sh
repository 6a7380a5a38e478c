#!/bin/bash
# N-rank bench rehearsal with the ranks sharing the one GPU of the box (not a measurement)
N=${1:-4}; shift
O=gpurun_out/r02_rehearse; mkdir -p $O
CCAL_P2P_TIMEOUT_MS=8000 timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus $N --steps 10 --warmup 2 --share-gpu "$@" > $O/n$N.out 2> $O/n$N.err
echo exit=$?; grep -v "^W1004\|^\[W" $O/n$N.err | tail -15 | cut -c1-400; tail -1 $O/n$N.out | cut -c1-3000
