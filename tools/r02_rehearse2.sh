#!/bin/bash
O=gpurun_out/r02_rehearse; mkdir -p $O
run() { tag=$1; shift; CCAL_P2P_TIMEOUT_MS=4000 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 4 --steps 10 --warmup 2 --share-gpu "$@" > $O/$tag.out 2> $O/$tag.err; echo "$tag exit=$?"; grep "\[bench\]" $O/$tag.err | head -2 | cut -c1-300; tail -1 $O/$tag.out | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['config']['allreduce'], d['ms_per_step'], d['lm'])
except Exception as e: print('no json', e)"; }
run c5_1600 --workload C5 --images 1600 --no-extras --timed-only
run c5_400 --workload C5 --images 400 --no-extras --timed-only
run c3_1600 --workload C3 --images 1600 --no-extras --timed-only
