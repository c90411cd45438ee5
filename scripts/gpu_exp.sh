#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
bash scripts/stall_hunt.sh 60 keep 2>&1 | tail -2
grep "stall-watch" gpurun_out/stall_hunt_keep.log | cut -c1-300
timeout -k 10 200 python -u scripts/debug_host_slow.py resnet50 256 2>&1 | tail -3
