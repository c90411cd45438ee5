#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 20 2>&1 | grep -v amdgpu.ids | cut -c1-92 || exit 1
for i in 1 2 3; do
  timeout -k 10 200 python -u bench.py 2>&1 | tail -1 | grep -o "ms_per_step\": [0-9.]*" | tr '\n' ' ' || exit 1
  echo
done
