#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 500 python -u benchmarks/kernel_bench.py --aten --json gpurun_out/kernel_bench.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kernel_bench.log
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench.log
