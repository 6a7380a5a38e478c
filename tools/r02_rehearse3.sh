#!/bin/bash
O=gpurun_out/r02_rehearse; mkdir -p $O
for N in 2 4; do
CCAL_P2P_TIMEOUT_MS=8000 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2957$N bench.py --gpus $N --steps 20 --warmup 3 --share-gpu > $O/share$N.out 2> $O/share$N.err
echo "N=$N exit=$?"; grep "\[bench\]" $O/share$N.err | head -3 | cut -c1-200
tail -1 $O/share$N.out | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print(d['config']['workload']); print(d['config']['allreduce'])
print('weak:', d.get('weak_scaling',{}).get('value'), d.get('weak_scaling',{}).get('config',{}).get('workload'))
print('c5:', d.get('c5',{}).get('value'), d.get('c5',{}).get('config',{}).get('workload'), d.get('c5',{}).get('config',{}).get('allreduce'))
print('roofline keys:', sorted(d['roofline']))
"
done
