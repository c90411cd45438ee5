#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3 || exit 1
for i in 1 2 3; do
  timeout -k 10 200 python -u bench.py 2>&1 | tail -1 | grep -o "ms_per_step\": [0-9.]*" | tr '\n' ' ' || exit 1
  echo
done
