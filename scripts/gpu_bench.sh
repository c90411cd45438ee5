#!/bin/bash
# tests + flagship bench + kernel-trace profile of the flagship step
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 400 python -u bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1 || { tail -30 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log; exit 1; }
echo profiled
