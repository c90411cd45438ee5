#!/bin/bash
# round-2 GPU session A: engine tests, full GPU suite, flagship bench with and without the forced exchange
OUT=gpurun_out/r2a
mkdir -p $OUT
step() {  # step <name> <timeout> <cmd...>: stop the session after a timeout / abort / segfault
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
step engine_tests 600 python -u -m pytest tests/test_gpu_engine.py -v --timeout 300 --timeout-method thread
tail -40 $OUT/engine_tests.log
step gpu_tests 600 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread --deselect tests/test_gpu_engine.py
tail -15 $OUT/gpu_tests.log
step bench 300 python bench.py --steps 20 --warmup 5
cat $OUT/bench.log; tail -3 $OUT/bench.err
HVD_FORCE_COLLECTIVES=1 step bench_forced 300 python bench.py --steps 20 --warmup 5
cat $OUT/bench_forced.log; tail -5 $OUT/bench_forced.err
exit 0
