#!/bin/bash
# row-streaming 3x3 forward / backward-data kernel: tests, isolated timing against the tile kernel, in-step A/B
set -o pipefail
OUT=gpurun_out/${TAG:-r3crows}
mkdir -p $OUT
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "conv_rows3x3 or conv_fwd_bwd" --timeout 120 --timeout-method thread 2>&1 | tail -15 || exit 1
DDL_CONV_ROWS=0 timeout -k 10 120 python -u benchmarks/conv_rows_bench.py || exit 1
DDL_CONV_ROWS=1 timeout -k 10 120 python -u benchmarks/conv_rows_bench.py || exit 1
if [ -n "$SWEEP" ]; then TAG=${TAG:-r3crows} bash scripts/gpu_r3_wg2.sh; fi
