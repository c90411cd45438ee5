#!/bin/bash
OUT=gpurun_out/r2m
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_numerics.py -x -q --timeout 300 --timeout-method thread -k "bottleneck or chain or residual or resnet50_trains or graphed or side_stream or direct or trajectory or linearity" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
grep -q " failed\|rror" $OUT/tests.log && { tail -60 $OUT/tests.log; exit 1; }
VARIANTS="X=1;DDL_LIN_GRAM_FWD=0" bash scripts/gpu_ab3.sh
