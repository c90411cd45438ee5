#!/bin/bash
# Rehearse the multi-rank bench path on a one-GPU box: (1) torchrun with one rank, (2) two ranks
# sharing the card over gloo (RCCL refuses two ranks on one device), small batch.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/dist_n1.log 2>&1 || { tail -30 gpurun_out/dist_n1.log; exit 1; }
tail -1 gpurun_out/dist_n1.log
HVD_BACKEND=gloo timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 > gpurun_out/dist_n2_gloo.log 2>&1 || { tail -40 gpurun_out/dist_n2_gloo.log; exit 1; }
tail -1 gpurun_out/dist_n2_gloo.log
