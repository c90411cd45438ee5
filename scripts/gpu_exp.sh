#!/bin/bash
set -o pipefail
for i in 1 2 3 4 5 6 7 8; do
timeout -k 10 200 python -u bench.py 2>&1 | tail -1 | grep -o "ms_per_step\": [0-9.]*" | tr '\n' ' ' || exit 1
echo
done
