#!/bin/bash
# phase cycles of several probe builds of the row-streaming conv kernel
set -o pipefail
for so in $SOS; do
  echo "== $so"
  DDL_KERNELS_SO=$PWD/benchmarks/native/$so timeout -k 10 120 python -u benchmarks/conv_rows_phases.py | grep "all  \|prologue\|wave [04]\|forward\|backward" || exit 1
done
