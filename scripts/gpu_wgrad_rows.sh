#!/bin/bash
# row-streaming 3x3 weight-gradient kernel: numerics, isolated, in-step (DDL_WGRAD_ROWS=0 is the tile kernel)
set -o pipefail
OUT=gpurun_out/${TAG:-wgrad_rows}
mkdir -p $OUT
timeout -k 10 500 python -u -m pytest tests/test_gpu_numerics.py tests/test_gpu_kernels.py -k "conv or block or chain or linearity or wgrad or resnet50_trains" -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/num.log 2>&1
tail -3 $OUT/num.log | cut -c1-300
for rows in 0 1; do
  DDL_WGRAD_ROWS=$rows timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 10 --only a3,b4,b2 > $OUT/kb$rows.log 2>&1 || { tail -20 $OUT/kb$rows.log; exit 1; }
  echo "rows=$rows: $(grep -E "^(a3|b4|b2) " $OUT/kb$rows.log | awk '{print $1, $8}' | tr '\n' ';')"
done
