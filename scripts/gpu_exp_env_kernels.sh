#!/bin/bash
# per-kernel time of chosen kernels (PATTERN, grep -E) under env variants: VARIANTS="A=1;B=2"
OUT=gpurun_out/envk
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
IFS=';' read -ra VS <<< "$VARIANTS"
i=0
for v in "${VS[@]}"; do
  i=$((i+1))
  (cd /tmp && env $v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/v$i -- python3 $R/bench.py --steps 6 --warmup 3 > $R/$OUT/v$i.log 2>&1)
  f=$(find $OUT/v$i -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "$PATTERN" $f | awk -F'",' '{print substr($1,1,70), $2}' | sed 's/"//g' | cut -d, -f1-3
  find $OUT/v$i -name "*kernel_trace.csv" -delete
done
