#!/bin/bash
# A/B of one environment switch over several models: VAR=<name> VALS="0 1" MODELS="resnet18 resnet34" [EXTRA="VAR2=V"]
set -o pipefail
OUT=gpurun_out/${TAG:-abm}
mkdir -p $OUT
for m in ${MODELS:-resnet50}; do
  for v in ${VALS:-0 1} ${VALS:-0 1}; do
    env $EXTRA $VAR=$v timeout -k 10 200 python -u bench.py --model $m --steps 30 --warmup 6 > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
    echo "$m $VAR=$v $EXTRA $(tail -1 $OUT/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])')" | tee -a $OUT/summary.txt
  done
done
