#!/bin/bash
set -o pipefail
DDL_CONV_GLDS=1 DDL_CONV_STAGES=3 timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3 || exit 1
for st in 2 3; do
  echo "== GLDS=1 STAGES=$st"
  DDL_CONV_GLDS=1 DDL_CONV_STAGES=$st timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 20 2>&1 | grep -v amdgpu.ids | cut -c1-62 || exit 1
done
for st in 2 3 2 3; do
  echo "== bench GLDS=1 STAGES=$st"
  DDL_CONV_GLDS=1 DDL_CONV_STAGES=$st timeout -k 10 200 python -u bench.py 2>&1 | tail -1 | grep -o "ms_per_step\": [0-9.]*" | tr '\n' ' ' || exit 1
  echo
done
