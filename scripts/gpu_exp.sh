#!/bin/bash
# phase probes of the conv main loop (library built with -DDDL_CONV_PROBES; results are wrong, timing only)
# bits: 1 no activation loads, 2 no weight loads, 4 no MFMA, 8 no LDS stores, 16 no LDS fragment reads, 32 no barrier
set -o pipefail
for pr in 0 3 7 11 19 15 27 23 31 35 63; do
  echo "== DDL_CONV_PROBE=$pr"
  DDL_CONV_BM=128 DDL_CONV_PROBE=$pr timeout -k 10 200 python -u benchmarks/kernel_bench.py --iters 20 --only c4,d1 2>&1 | grep "^c4\|^d1" | cut -c1-60 || exit 1
done
