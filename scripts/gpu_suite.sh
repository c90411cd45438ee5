#!/bin/bash
# whole GPU test suite (one pytest process), then smoke(), then the flagship bench
set -o pipefail
OUT=gpurun_out/${TAG:-suite}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 ${SUITE_TIMEOUT:-1000} python -u -m pytest tests -m gpu -x -v --timeout 240 --timeout-method thread ${PYTEST_ARGS} > $OUT/pytest_gpu.log 2>&1
rc=$?
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
if [ $rc -ne 0 ]; then grep -nE "FAILED|Error|assert" $OUT/pytest_gpu.log | tail -40; tail -60 $OUT/pytest_gpu.log; exit $rc; fi
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -30 $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 > $OUT/bench.log 2>&1 || { tail -30 $OUT/bench.log; exit 1; }
tail -1 $OUT/bench.log
