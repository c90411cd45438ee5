#!/bin/bash
for c in 268435456 1099511627776 134217728 67108864 0; do
  echo "== DDL_NT_BYTES=$c"
  DDL_NT_BYTES=$c timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c95-230 || exit 1
done
