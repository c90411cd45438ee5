#!/bin/bash
# stock PyTorch-ROCm yard-stick with MIOpen find mode (cudnn.benchmark=True); long and silent, so
# the output goes straight to a file that keeps growing (heartbeat lines)
mkdir -p gpurun_out
timeout -k 10 1000 python -u benchmarks/stock_pytorch.py --steps 20 --warmup 8 --batch 256 --autotune > gpurun_out/stock_autotune.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/stock_autotune.log
timeout -k 10 1000 python -u benchmarks/stock_pytorch.py --steps 20 --warmup 8 --batch 256 --pure-bf16 --autotune > gpurun_out/stock_autotune_pure.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/stock_autotune_pure.log
