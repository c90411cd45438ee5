#!/bin/bash
OUT=gpurun_out/r2b
mkdir -p $OUT
step() {
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
W="env DDL_RCCL_SKIP_COMM=1 HVD_ALLREDUCE_ALGO=twoshot HVD_COLLECTIVE_TIMEOUT_S=60 DDL_COMM_STREAM_PRIORITY=low python -m distributeddeeplearning_amd.launch.run"
TEST_STEP_MODE=sync step dp_sync 200 $W -np 2 python -u tests/gpu_worker.py dp_equivalence
grep -h "RESULT\|ddl xgmi\|ERROR" $OUT/dp_sync.log | sed 's/.*"trace"/trace/' | cut -c1-900
step dp_step 200 $W -np 2 python -u tests/gpu_worker.py dp_equivalence
grep -h "RESULT\|ddl xgmi\|ERROR" $OUT/dp_step.log | sed 's/.*"trace"/trace/' | cut -c1-900
HVD_PER_BUCKET_UPDATE=0 step dp_nobucket 200 $W -np 2 python -u tests/gpu_worker.py dp_equivalence
grep -h "RESULT\|ddl xgmi\|ERROR" $OUT/dp_nobucket.log | sed 's/.*"trace"/trace/' | cut -c1-900
exit 0
