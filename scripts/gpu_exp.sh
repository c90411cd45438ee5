#!/bin/bash
mkdir -p gpurun_out
for x in 0 20000 45000; do
  echo "== DDL_CONV_EXTRA_LDS=$x"
  DDL_CONV_EXTRA_LDS=$x timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 20 --only c3,c4,c5,d1,d4,b4,a3,a4 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
done
