#!/bin/bash
# conv_igemm with an LDS-DMA ring (probe library built with -DDDL_CONV_STAGES=3) against the default library
set -o pipefail
OUT=gpurun_out/${TAG:-r3ring}
mkdir -p $OUT
P=$GRAFT_REPO_ROOT/benchmarks/native/_ddl_kernels_probe.so
for lib in "" $P; do
  DDL_KERNELS_SO=$lib timeout -k 10 500 python -u -m pytest tests/test_gpu_numerics.py tests/test_gpu_kernels.py -k "conv or block or chain or linearity or resnet50_trains" -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/num.log 2>&1
  echo "lib=${lib:-default}: $(tail -1 $OUT/num.log | cut -c1-200)"
  DDL_KERNELS_SO=$lib timeout -k 10 300 python -u benchmarks/kernel_bench.py --iters 10 > $OUT/kb.log 2>&1 || { tail -20 $OUT/kb.log; exit 1; }
  grep -E "^[a-d][1-6] " $OUT/kb.log | awk '{print $1, $6, $7}' | tr '\n' ';'; echo
  tail -1 $OUT/kb.log
done
for lib in "" $P "" $P; do
  DDL_KERNELS_SO=$lib timeout -k 10 200 python -u bench.py --steps 30 --warmup 6 > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
  echo "step lib=${lib:-default} $(tail -1 $OUT/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])')"
done
