#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "graphed or fused_sgd or lars" --timeout 200 --timeout-method thread 2>&1 | tail -30 || exit 1
timeout -k 10 200 python -u bench.py --graph 2>&1 | tail -3 || exit 1
