#!/bin/bash
# EPI-4 (block-input gradient) in isolation: operand loads dropped / fewer workgroups per CU
OUT=gpurun_out/epi4
mkdir -p $OUT
for v in "X=0" "DDL_CONV_PROBE=3" "DDL_CONV_EXTRA_LDS=14000" "DDL_CONV_EXTRA_LDS=45000"; do
  env $v timeout -k 10 300 python benchmarks/kernel_bench.py --iters 20 --epi4 --only a2,b3 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  echo "== $v"; grep -v "amdgpu.ids" $OUT/run.log | cut -c1-150 | grep -v "per-step\|^id"
done
