#!/bin/bash
mkdir -p gpurun_out
for p in low high; do
  echo "== DDL_WGRAD_STREAM_PRIORITY=$p"
  DDL_WGRAD_STREAM_PRIORITY=$p timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c95-260 || exit 1
  DDL_WGRAD_STREAM_PRIORITY=$p HVD_FORCE_COLLECTIVES=1 timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c95-260 || exit 1
done
