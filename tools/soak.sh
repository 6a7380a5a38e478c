#!/bin/bash
# p2p soak: 2 and 4 ranks sharing the one GPU repeat a sharded fit and require bit-identical results (tools/p2p_soak.py)
O=gpurun_out/soak; mkdir -p $O
export CCAL_P2P_TIMEOUT_MS=4000
for r in 2 4; do
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $r --master-addr 127.0.0.1 --master-port $((29610 + r)) tools/p2p_soak.py --solves ${SOLVES:-30000} > $O/soak$r.log 2>&1 || { echo "soak $r failed"; tail -5 $O/soak$r.log; exit 1; }
  grep "solves of" $O/soak$r.log
done
