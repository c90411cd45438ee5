#!/bin/bash
# round-2 GPU session E: transposer on the side stream, KRSC dgrad A/B, >2 GiB kernels, large batches
OUT=gpurun_out/r2e
mkdir -p $OUT
step() {
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
step kernels 900 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 --timeout-method thread
tail -5 $OUT/kernels.log
[ "$(grep -c ' passed' $OUT/kernels.log)" = "0" ] && exit 1
DDL_DGRAD_KRSC=1 step kernels_krsc 900 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 --timeout-method thread
tail -3 $OUT/kernels_krsc.log
step numerics 900 python -u -m pytest tests/test_gpu_numerics.py -x -q --timeout 600 --timeout-method thread -k "beyond or 1408"
tail -25 $OUT/numerics.log
for i in 1 2; do
step bench$i 300 python bench.py --steps 20 --warmup 5
cut -c1-330 $OUT/bench$i.log
DDL_DGRAD_KRSC=1 step bench_krsc$i 300 python bench.py --steps 20 --warmup 5
cut -c1-330 $OUT/bench_krsc$i.log
done
step b1024 400 python bench.py --steps 8 --warmup 3 --batch 1024 --optimizer lars
cut -c1-500 $OUT/b1024.log; tail -3 $OUT/b1024.err
step b2048 500 python bench.py --steps 6 --warmup 2 --batch 2048 --optimizer lars
cut -c1-900 $OUT/b2048.log; tail -3 $OUT/b2048.err
exit 0
