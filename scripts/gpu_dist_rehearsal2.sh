#!/bin/bash
# Multi-rank bench path on a one-GPU box, final tree: torchrun with one rank (native engine + RCCL), two ranks over
# gloo, two ranks through the native engine's direct xGMI kernel (ranks share the card: RCCL itself refuses that)
set -o pipefail
OUT=gpurun_out/${TAG:-dist2}
mkdir -p $OUT
timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 > $OUT/tr_n1.log 2>&1 || { tail -30 $OUT/tr_n1.log; exit 1; }
grep metric $OUT/tr_n1.log | cut -c1-400
HVD_BACKEND=gloo timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 > $OUT/tr_n2_gloo.log 2>&1 || { tail -40 $OUT/tr_n2_gloo.log; exit 1; }
grep metric $OUT/tr_n2_gloo.log | cut -c1-400
DDL_RCCL_SKIP_COMM=1 HVD_ALLREDUCE_ALGO=twoshot DDL_COMM_STREAM_PRIORITY=low HVD_COLLECTIVE_TIMEOUT_S=60 \
  timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 > $OUT/tr_n2_xgmi.log 2>&1 || { tail -40 $OUT/tr_n2_xgmi.log; exit 1; }
grep metric $OUT/tr_n2_xgmi.log | cut -c1-1400
