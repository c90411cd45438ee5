#!/bin/bash
# row-streaming conv kernel: tests, phase cycles (probe library) and kernel durations from a kernel trace
set -o pipefail
OUT=gpurun_out/${TAG:-conv_rows}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "conv_rows3x3 or conv_fwd_bwd" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
if [ -f benchmarks/native/_ddl_kernels_probe.so ] && [ -z "$NOPROBE" ]; then
DDL_KERNELS_SO=$PWD/benchmarks/native/_ddl_kernels_probe.so timeout -k 10 120 python -u benchmarks/conv_rows_phases.py || exit 1
fi
export TMPDIR=/tmp
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o rows -- python3 $R/benchmarks/conv_rows_bench.py > $R/$OUT/prof.log 2>&1 || { tail -20 $R/$OUT/prof.log; exit 1; }
grep -h "conv_rows\|DDL_CONV" $R/$OUT/prof.log $R/$OUT/rows_kernel_stats.csv | cut -c1-200
