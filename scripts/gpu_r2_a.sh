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
step engine_tests 900 python -u -m pytest tests/test_gpu_engine.py -v -s --timeout 300 --timeout-method thread
tail -40 $OUT/engine_tests.log
step gpu_tests 600 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread --deselect tests/test_gpu_engine.py
tail -15 $OUT/gpu_tests.log
step bench 300 python bench.py --steps 20 --warmup 5
cat $OUT/bench.log; tail -3 $OUT/bench.err
HVD_FORCE_COLLECTIVES=1 step bench_forced 300 python bench.py --steps 20 --warmup 5
cat $OUT/bench_forced.log; tail -5 $OUT/bench_forced.err
# kernel trace of the forced-exchange step: the RCCL kernel must show up (launched by the engine's comm thread)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && HVD_FORCE_COLLECTIVES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_forced -- python3 $R/bench.py --steps 6 --warmup 3 > $R/$OUT/prof_forced.log 2>&1)
echo "prof_forced rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_forced -name "*kernel_stats.csv" | head -1 | xargs -r head -40
find $OUT/prof_forced -name "*kernel_trace.csv" -size +30M -delete
exit 0
