#!/bin/bash
# round-2 GPU session H: BatchNorm backward of the wide unit by linearity: numerics, then A/B bench
OUT=gpurun_out/r2h
mkdir -p $OUT
step() {
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
step blocks 600 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 --timeout-method thread -k "bottleneck or chain or residual or resnet50_trains or graphed or side_stream"
tail -30 $OUT/blocks.log | cut -c1-300
grep -q " failed" $OUT/blocks.log && exit 1
step traj 600 python -u -m pytest tests/test_gpu_numerics.py -x -q -s --timeout 300 --timeout-method thread -k "trajectory"
grep -E "resnet|passed|failed" $OUT/traj.log | cut -c1-400
for i in 1 2; do
step bench_lin$i 300 python bench.py --steps 20 --warmup 5
cut -c1-330 $OUT/bench_lin$i.log
DDL_BN3_LINEAR=0 step bench_old$i 300 python bench.py --steps 20 --warmup 5
cut -c1-330 $OUT/bench_old$i.log
done
exit 0
