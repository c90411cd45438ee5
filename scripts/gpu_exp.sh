#!/bin/bash
set -o pipefail
# host-bound configuration: the GPU work of a 16-image batch is far shorter than the launch sequence
timeout -k 10 200 python -u bench.py --batch 16 2>&1 | tail -1 | grep -o "ms_per_step\": [0-9.]*" || exit 1
timeout -k 10 200 python -u bench.py --steps 6 --warmup 8 2>&1 | tail -1 | grep -o "ms_per_step\": [0-9.]*" || exit 1
