#!/bin/bash
# gemm256.hip: numerics, then the wide 1x1 layers in isolation with and without the kernel (and against hipBLASLt)
set -o pipefail
OUT=gpurun_out/${TAG:-gemm256}
mkdir -p $OUT
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "conv_fwd_bwd or wide_plain_gemm" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
ONLY=${ONLY:-c1,c3,c5,d1,d3,d5}
DDL_GEMM256=0 timeout -k 10 200 python -u benchmarks/kernel_bench.py --only $ONLY --blaslt --iters 20 > $OUT/kb_off.log 2>&1 || { tail -20 $OUT/kb_off.log; exit 1; }
for cfg in "DDL_GEMM256_NST=3" "DDL_GEMM256_NST=4"; do
  env $cfg DDL_GEMM256_MIN_TILES=0 timeout -k 10 200 python -u benchmarks/kernel_bench.py --only $ONLY --iters 20 > $OUT/kb_$cfg.log 2>&1 || { tail -20 $OUT/kb_$cfg.log; exit 1; }
done
for f in $OUT/kb_*.log; do echo "== $f"; cut -c1-120 $f | grep -v "^id\|per-step"; done
grep -o "hipBLASLt gemm.*TF/s" $OUT/kb_off.log
