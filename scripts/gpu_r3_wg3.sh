#!/bin/bash
set -o pipefail
OUT=gpurun_out/${TAG:-r3wg3}
mkdir -p $OUT
DDL_WGRAD_BIG_PIPE=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_numerics.py -k "conv_resnet50_shape or beyond_2gib" -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/num.log 2>&1
tail -2 $OUT/num.log | cut -c1-300
for pipe in 0 1; do
  DDL_WGRAD_BIG_TARGET=256 DDL_WGRAD_BIG_PIPE=$pipe timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 10 --only b6,c1,c2,c3,c4,c5,c6,d1,d2,d3,d4,d5,d6 > $OUT/kb_pipe$pipe.log 2>&1 || { tail -20 $OUT/kb_pipe$pipe.log; exit 1; }
  echo "pipe=$pipe (256 workgroups)"; grep -E "^(b6|c[1-6]|d[1-6]) " $OUT/kb_pipe$pipe.log | awk '{print $1, $8}' | tr '\n' ';'; echo
done
