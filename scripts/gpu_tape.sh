#!/bin/bash
# tape tests, then the flagship eagerly / from the tape, then the kernel statistics of the taped run
set -o pipefail
OUT=gpurun_out/${TAG:-tape}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_tape.py tests/test_gpu_kernels.py -k "tape or subsample" -m gpu -v --timeout 240 --timeout-method thread > $OUT/tape.log 2>&1
grep -E "PASSED|FAILED|TapeError|first seen|passed|failed" $OUT/tape.log | cut -c1-400
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --tape off > $OUT/bench_eager.log 2>&1 || { tail -20 $OUT/bench_eager.log; exit 1; }
tail -1 $OUT/bench_eager.log | cut -c1-330
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --tape on > $OUT/bench_tape.log 2>&1 || { tail -20 $OUT/bench_tape.log; exit 1; }
tail -1 $OUT/bench_tape.log | cut -c1-330
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 6 --tape on > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$OUT/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT && ls -R $OUT/prof | head -20
