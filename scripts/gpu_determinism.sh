#!/bin/bash
# run-to-run determinism screen of the training step: N identical bench runs, final losses counted
# MODEL=resnet101 N=24 [ENVS="VAR=V ..."] [STEPS=30] [ARGS="--tape off"] bash scripts/gpu_determinism.sh
set -o pipefail
OUT=gpurun_out/${TAG:-det}
mkdir -p $OUT
: > $OUT/losses.txt
for i in $(seq 1 ${N:-24}); do
  env $ENVS timeout -k 10 200 python -u bench.py --model ${MODEL:-resnet101} --steps ${STEPS:-30} --warmup 6 $ARGS > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
  tail -1 $OUT/bench.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["final_loss"], d["ms_per_step"])' | tee -a $OUT/losses.txt
done
echo "== ${MODEL:-resnet101} $ENVS: distinct final losses"
cut -d" " -f1 $OUT/losses.txt | sort | uniq -c
