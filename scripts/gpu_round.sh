#!/bin/bash
# First-contact GPU run: smoke, flagship bench, stock-PyTorch yard-stick, kernel-trace profile.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -30 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 400 python -u bench.py --steps 20 --warmup 8 > gpurun_out/bench.log 2>&1 || { tail -30 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
timeout -k 10 400 python -u benchmarks/stock_pytorch.py --steps 20 --warmup 8 --batch 256 > gpurun_out/stock.log 2>&1 || { tail -30 gpurun_out/stock.log; exit 1; }
tail -1 gpurun_out/stock.log
timeout -k 10 400 python -u benchmarks/stock_pytorch.py --steps 20 --warmup 8 --batch 256 --pure-bf16 > gpurun_out/stock_pure.log 2>&1 || { tail -30 gpurun_out/stock_pure.log; exit 1; }
tail -1 gpurun_out/stock_pure.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log; exit 1; }
echo done
