#!/bin/bash
# kernel trace of the taped flagship under an environment given in ENVS (space separated VAR=VAL), output tag TAG
set -o pipefail
OUT=gpurun_out/${TAG:-trace}
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && env $ENVS timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o bench -- python3 $R/bench.py --steps 10 --warmup 6 --tape on > $R/$OUT/prof.log 2>&1 || { tail -20 $R/$OUT/prof.log; exit 1; }
tail -1 $R/$OUT/prof.log | cut -c1-200
