#!/bin/bash
OUT=gpurun_out/r2h2
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 --timeout-method thread -k "bottleneck or chain or residual or resnet50_trains or graphed or side_stream" > $OUT/blocks.log 2>&1
tail -3 $OUT/blocks.log
grep -q " failed\|rror" $OUT/blocks.log && { tail -40 $OUT/blocks.log; exit 1; }
AB_ENV="DDL_BN3_LINEAR=0" bash scripts/gpu_ab.sh
