#!/bin/bash
# Rehearse bench.py's N-rank path on a 1-GPU box (ranks share the GPU, gloo): checks sharding, gather, assembly, JSON.
set -e
N=${1:-2}
RT_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus $N --steps 2 --warmup 1
