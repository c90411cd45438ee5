#!/bin/bash
# secondary configurations on one GPU: other depths, larger per-GPU batches, LARS, forced exchange
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/configs.log
: > $out
run() {
  echo "== $*" | tee -a $out
  timeout -k 10 300 "$@" > gpurun_out/configs_last.log 2>&1 || { tail -30 gpurun_out/configs_last.log; exit 1; }
  tail -1 gpurun_out/configs_last.log | tee -a $out
}
run python -u bench.py
run python -u bench.py --optimizer lars
run python -u bench.py --batch 512 --steps 16 --warmup 4
run python -u bench.py --batch 1024 --steps 8 --warmup 3
run python -u bench.py --batch 1024 --optimizer lars --steps 8 --warmup 3
run python -u bench.py --model resnet18
run python -u bench.py --model resnet34
run python -u bench.py --model resnet101 --steps 15 --warmup 5
run python -u bench.py --model resnet152 --steps 15 --warmup 5
run python -u bench.py --model resnet200 --steps 10 --warmup 4
run env HVD_FORCE_COLLECTIVES=1 python -u bench.py
run env HVD_FORCE_COLLECTIVES=1 python -u bench.py --model resnet152 --steps 15 --warmup 5
