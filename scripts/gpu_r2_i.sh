#!/bin/bash
OUT=gpurun_out/r2i
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_numerics.py -x -q --timeout 300 --timeout-method thread -k "not beyond and not 1408 and not batch_256" > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
grep -q " failed\|rror" $OUT/tests.log && { tail -60 $OUT/tests.log; exit 1; }
AB_ENV="DDL_BN_ONE_LAUNCH=0" bash scripts/gpu_ab.sh
