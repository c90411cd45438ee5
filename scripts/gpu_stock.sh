#!/bin/bash
# stock PyTorch-ROCm yard-stick in its four variants
mkdir -p gpurun_out
for mode in "" "--pure-bf16" "--autotune" "--pure-bf16 --autotune"; do
  timeout -k 10 500 python -u benchmarks/stock_pytorch.py --steps 20 --warmup 8 --batch 256 $mode 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/stock.log | tail -1
done
