#!/bin/bash
# in-step sweep: SWEEP="VAR=V,VAR2=W VAR=X ..." runs the flagship once per space-separated entry (commas join several variables)
set -o pipefail
OUT=gpurun_out/${TAG:-sweep}
mkdir -p $OUT
run() {
  env "$@" timeout -k 10 200 python -u bench.py --steps 30 --warmup 6 > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
  echo "step $* $(tail -1 $OUT/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])')" | tee -a $OUT/summary.txt
}
for spec in $SWEEP; do
  run $(echo $spec | tr ',' ' ')
done
