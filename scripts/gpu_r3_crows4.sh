#!/bin/bash
# numerics suite + tape tests + in-step A/B on the tree with the row-streaming conv kernel
set -o pipefail
OUT=gpurun_out/${TAG:-r3crows4}
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_gpu_numerics.py tests/test_gpu_tape.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5 || exit 1
SWEEP="DDL_CONV_ROWS=0 DDL_CONV_ROWS=1 DDL_CONV_ROWS=0 DDL_CONV_ROWS=1" TAG=${TAG:-r3crows4} bash scripts/gpu_r3_wg2.sh
