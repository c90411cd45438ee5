#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ds -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_ds.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_ds.log; exit 1; }
echo profiled
