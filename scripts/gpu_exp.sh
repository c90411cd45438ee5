#!/bin/bash
for t in 134217728 33554432 16777216 8388608 4194304; do
  echo "== HVD_FUSION_THRESHOLD=$t"
  HVD_FORCE_COLLECTIVES=1 HVD_FUSION_THRESHOLD=$t HVD_FIRST_BUCKET_BYTES=$t timeout -k 10 200 python -u bench.py 2>&1 | tail -1 | grep -o "ms_per_step\": [0-9.]*" | tr '\n' ' ' || exit 1
  echo
done
