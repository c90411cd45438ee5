#!/bin/bash
# secondary BASELINE configs on one GPU: large per-GPU batch (LARS / SGD), ResNet-152, forced exchange
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
run python -u bench.py --batch 1024 --optimizer lars --steps 8 --warmup 3
run python -u bench.py --batch 1024 --steps 8 --warmup 3
run python -u bench.py --model resnet152 --steps 12 --warmup 4
run python -u bench.py --optimizer lars --steps 20 --warmup 5
run env HVD_FORCE_COLLECTIVES=1 python -u bench.py --model resnet152 --steps 12 --warmup 4
rocm-smi --showmeminfo vram 2>/dev/null | tail -4 | tee -a $out
