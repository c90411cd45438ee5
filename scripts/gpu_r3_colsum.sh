#!/bin/bash
set -o pipefail
OUT=gpurun_out/${TAG:-r3colsum}
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_numerics.py -x -q -k "column_sums or lin or block or traject or residual or reproducible" --timeout 200 --timeout-method thread 2>&1 | tail -4 || exit 1
SWEEP="DDL_WGRAD_COLSUM=0 DDL_WGRAD_COLSUM=1 DDL_WGRAD_COLSUM=0 DDL_WGRAD_COLSUM=1" TAG=${TAG:-r3colsum} bash scripts/gpu_r3_wg2.sh
