#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --optimizer lars 2>&1 | tail -1 | tee gpurun_out/bench_lars.log || exit 1
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --model resnet152 2>&1 | tail -1 | tee gpurun_out/bench_r152.log || exit 1
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --model resnet101 2>&1 | tail -1 | tee gpurun_out/bench_r101.log || exit 1
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --model resnet18 2>&1 | tail -1 | tee gpurun_out/bench_r18.log || exit 1
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --batch 512 2>&1 | tail -1 | tee gpurun_out/bench_b512.log || exit 1
