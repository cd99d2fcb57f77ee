#!/bin/bash
# Rehearsal of `bench.py --mode stream --gpus 2` on a ONE-GPU box: both ranks on cuda:0, gather staged through
# the host under gloo (RCCL refuses two ranks on one device).  Control flow only - not a measurement.
BAS_BENCH_ONE_DEVICE=1 BAS_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --mode stream --gpus 2 --steps 3 --warmup 1 --sources 64 --block 65536
