#!/bin/bash
# kernel trace of the flagship step with the one-rank RCCL exchange forced on
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
export HVD_FORCE_COLLECTIVES=1
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_forced -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_forced.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_forced.log; exit 1; }
echo profiled
