#!/bin/bash
OUT=gpurun_out/r2k
mkdir -p $OUT
timeout -k 10 1100 python -u -m pytest tests -m gpu -q ${SEL:+-k "$SEL"} --timeout 600 --timeout-method thread > $OUT/gpu_tests.log 2>&1
echo "rc=$?"
tail -30 $OUT/gpu_tests.log | cut -c1-300
