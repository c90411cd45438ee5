#!/bin/bash
# interleaved reduction order of the plain-GEMM weight gradients: conv tests (default and forced for every plain GEMM), then the in-step A/B
set -o pipefail
OUT=gpurun_out/${TAG:-r3il}
mkdir -p $OUT
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "conv_fwd_bwd or side_stream" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
DDL_WGRAD_INTERLEAVE=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "conv_fwd_bwd or side_stream" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
SWEEP="${SWEEP:-DDL_WGRAD_INTERLEAVE=0 DDL_WGRAD_INTERLEAVE=100000 DDL_WGRAD_INTERLEAVE=0 DDL_WGRAD_INTERLEAVE=100000}" TAG=${TAG:-r3il} bash scripts/gpu_r3_wg2.sh
