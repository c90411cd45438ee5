#!/bin/bash
# round-2 GPU session D: re-run of selected tests
OUT=gpurun_out/r2d
mkdir -p $OUT
step() {
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
step sel 900 python -u -m pytest tests/test_gpu_numerics.py tests/test_gpu_engine.py -v -s --timeout 300 --timeout-method thread -k "${SEL:-trajectory or validate or three_processes}"
grep -E "PASSED|FAILED|ERROR|resnet.. (fp32|hip)|passed|failed|RESULT" $OUT/sel.log | cut -c1-700
grep -B5 -A25 "^E  " $OUT/sel.log | head -80
exit 0
