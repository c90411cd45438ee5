#!/bin/bash
# kernel trace + stats of the flagship step (current build)
OUT=gpurun_out/r2prof
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/step -- python3 $R/bench.py --steps 6 --warmup 3 > $R/$OUT/step.log 2>&1)
echo "prof rc=$?"
f=$(find $OUT/step -name "*kernel_stats.csv" | head -1)
head -45 $f | cut -c1-230
t=$(find $OUT/step -name "*kernel_trace.csv" | head -1)
python scripts/trace_gaps.py $t | head -30
