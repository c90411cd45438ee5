#!/bin/bash
mkdir -p gpurun_out
for t in 256 384 640 896 3072 4096; do
  echo "== DDL_WGRAD_TARGET=$t"
  DDL_WGRAD_TARGET=$t timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 20 2>&1 | grep -v amdgpu.ids > gpurun_out/kb_wt$t.log || exit 1
  awk '{printf "%s:%s ", $1, $8}' gpurun_out/kb_wt$t.log; echo; tail -1 gpurun_out/kb_wt$t.log
done
