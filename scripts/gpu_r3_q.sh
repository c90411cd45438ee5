#!/bin/bash
# one-launch weight preparation of the linear path: kernel test, block / trajectory tests, in-step A/B
set -o pipefail
OUT=gpurun_out/${TAG:-r3q}
mkdir -p $OUT
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_numerics.py -x -q -k "lin or block or traject or residual" --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
SWEEP="${SWEEP:-DDL_LIN_PREP_Q=0 DDL_LIN_PREP_Q=1 DDL_LIN_PREP_Q=0 DDL_LIN_PREP_Q=1}" TAG=${TAG:-r3q} bash scripts/gpu_r3_wg2.sh
