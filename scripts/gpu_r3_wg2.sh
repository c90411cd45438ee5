#!/bin/bash
# in-step sweep of the big weight-gradient kernel's knobs (alternating with the baseline)
set -o pipefail
OUT=gpurun_out/${TAG:-r3wg2}
mkdir -p $OUT
run() {
  env "$@" timeout -k 10 200 python -u bench.py --steps 30 --warmup 6 > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
  echo "step $* $(tail -1 $OUT/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])')"
}
for spec in $SWEEP; do
  run $(echo $spec | tr ',' ' ')
done
