#!/bin/bash
# > 2 GiB two-block test, late-peer cost table, torchrun rehearsals (1 rank; 2 ranks on one GPU over gloo and over the xGMI kernel)
set -o pipefail
OUT=gpurun_out/${TAG:-r3b}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 500 python -u -m pytest tests/test_gpu_numerics.py -k "two_layer1_blocks" -m gpu -v --timeout 400 --timeout-method thread > $OUT/big2.log 2>&1
grep -E "PASSED|FAILED|Error|assert|passed|failed" $OUT/big2.log | cut -c1-300 | head
timeout -k 10 600 python -u benchmarks/late_peer.py --json $OUT/late_peer.json > $OUT/late_peer.log 2>&1 || { tail -30 $OUT/late_peer.log; exit 1; }
tail -8 $OUT/late_peer.log
timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 4 > $OUT/dist_n1.log 2>&1 || { tail -30 $OUT/dist_n1.log; exit 1; }
tail -1 $OUT/dist_n1.log | cut -c1-400
HVD_BACKEND=gloo timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 > $OUT/dist_n2_gloo.log 2>&1 || { tail -40 $OUT/dist_n2_gloo.log; exit 1; }
tail -1 $OUT/dist_n2_gloo.log | cut -c1-400
DDL_RCCL_SKIP_COMM=1 HVD_ALLREDUCE_ALGO=twoshot DDL_COMM_STREAM_PRIORITY=low timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 > $OUT/dist_n2_xgmi.log 2>&1 || { tail -40 $OUT/dist_n2_xgmi.log; exit 1; }
tail -1 $OUT/dist_n2_xgmi.log
