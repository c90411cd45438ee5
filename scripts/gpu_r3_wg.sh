#!/bin/bash
# the 256x256 weight-gradient kernel: numerics at the production shapes, then isolated and in-step timing
set -o pipefail
OUT=gpurun_out/${TAG:-r3wg}
mkdir -p $OUT
DDL_WGRAD_BIG=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_numerics.py -k "conv_resnet50_shape or beyond_2gib" -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/num.log 2>&1
tail -5 $OUT/num.log | cut -c1-300
for big in 0 1; do
  DDL_WGRAD_BIG=$big timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 10 --only b6,c1,c2,c3,c4,c5,c6,d1,d2,d3,d4,d5,d6 > $OUT/kb_big$big.log 2>&1 || { tail -20 $OUT/kb_big$big.log; exit 1; }
  echo "big=$big"; grep -E "^(b6|c[1-6]|d[1-6]) " $OUT/kb_big$big.log | awk '{print $1, $2, $3, $8, $14}' | tr '\n' ';'; echo
done
for cfg in "0" "1" "1 DDL_WGRAD_BIG_NST=4" "1 DDL_WGRAD_BIG_TARGET=128" "0"; do
  set -- $cfg
  env DDL_WGRAD_BIG=$1 $2 timeout -k 10 200 python -u bench.py --steps 20 --warmup 6 > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
  echo "step big=$cfg $(tail -1 $OUT/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])')"
done
