#!/bin/bash
# round-2 GPU session C: new numerics tests, whole GPU suite, smoke, examples
OUT=gpurun_out/r2c
mkdir -p $OUT
step() {
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
step numerics 900 python -u -m pytest tests/test_gpu_numerics.py -v -s --timeout 300 --timeout-method thread
grep -E "PASSED|FAILED|ERROR|resnet.. (fp32|hip)|passed|failed" $OUT/numerics.log | cut -c1-400
grep -B5 -A25 "^E  " $OUT/numerics.log | head -150
step gpu_tests 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread --deselect tests/test_gpu_numerics.py
tail -30 $OUT/gpu_tests.log
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
tail -3 $OUT/smoke.log; tail -5 $OUT/smoke.err
exit 0
