#!/bin/bash
# old (128x128) weight-gradient kernel with the DMA hidden from the compiler: numerics, isolated, in-step (old library = baseline)
set -o pipefail
OUT=gpurun_out/${TAG:-r3wg5}
mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_numerics.py tests/test_gpu_kernels.py -k "conv_resnet50_shape or beyond_2gib or conv_fwd_bwd or wgrad" -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/num.log 2>&1
tail -2 $OUT/num.log | cut -c1-300
timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 10 > $OUT/kb.log 2>&1 || { tail -20 $OUT/kb.log; exit 1; }
grep -E "^[a-d][1-6] " $OUT/kb.log | awk '{print $1, $8}' | tr '\n' ';'; echo
tail -1 $OUT/kb.log
