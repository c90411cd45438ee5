#!/bin/bash
mkdir -p gpurun_out
for bk in 64 32; do
  echo "== DDL_CONV_BK=$bk"
  DDL_CONV_BK=$bk timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kb_bk$bk.log | tail -24
done
