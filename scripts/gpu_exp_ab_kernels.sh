#!/bin/bash
# per-kernel time of chosen kernels (PATTERN, grep -E): current library against benchmarks/ab/_ddl_kernels_old.so,
# then the training step alternated between the two
OUT=gpurun_out/abk
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in new old; do
  if [ $v = old ]; then export DDL_KERNELS_SO=$R/benchmarks/ab/_ddl_kernels_old.so; else unset DDL_KERNELS_SO; fi
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/$v -- python3 $R/bench.py --steps 6 --warmup 3 > $R/$OUT/$v.log 2>&1)
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "$PATTERN" $f | awk -F'",' '{print substr($1,1,70), $2}' | sed 's/"//g' | cut -d, -f1-3
  find $OUT/$v -name "*kernel_trace.csv" -delete
done
unset DDL_KERNELS_SO
VARIANTS="X=0;DDL_KERNELS_SO=$R/benchmarks/ab/_ddl_kernels_old.so" bash scripts/gpu_ab3.sh
