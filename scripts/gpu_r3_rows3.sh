#!/bin/bash
set -o pipefail
OUT=gpurun_out/${TAG:-r3rows3}
mkdir -p $OUT
timeout -k 10 500 python -u -m pytest tests/test_gpu_numerics.py tests/test_gpu_kernels.py -k "conv or wgrad or resnet50_trains" -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/num.log 2>&1
tail -3 $OUT/num.log | cut -c1-300
for v in 0 1 0 1; do
  DDL_WGRAD_ROWS_SKEW=$v timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 20 --only a3,b4 > $OUT/kb$v.log 2>&1 || { tail -20 $OUT/kb$v.log; exit 1; }
  echo "skew=$v: $(grep -E "^(a3|b4|b2) " $OUT/kb$v.log | awk '{print $1, $8}' | tr '\n' ';')"
done
