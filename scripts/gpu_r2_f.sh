#!/bin/bash
OUT=gpurun_out/r2f
mkdir -p $OUT
step() {
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
step gpu_tests 1100 python -u -m pytest tests -m gpu -q --timeout 600 --timeout-method thread
tail -12 $OUT/gpu_tests.log
exit 0
