#!/bin/bash
# kernel trace of the flagship step, analysed by phase (forward / backward / update)
OUT=gpurun_out/r2phases
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/step -- python3 $R/bench.py --steps 6 --warmup 3 > $R/$OUT/step.log 2>&1)
echo "prof rc=$?"
t=$(find $OUT/step -name "*kernel_trace.csv" | head -1)
python scripts/trace_phases.py $t > $OUT/phases.txt
python scripts/trace_phases.py $t --list > $OUT/list.txt
cat $OUT/phases.txt
rm -rf $OUT/step
