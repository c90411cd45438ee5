#!/bin/bash
# run-to-run determinism screen of the training step: N identical bench runs, final losses counted
# MODEL=resnet101 N=24 [ENVS="VAR=V ..."] [STEPS=30] [ARGS="--tape off"] [RANKS=2] bash scripts/gpu_determinism.sh
set -o pipefail
OUT=gpurun_out/${TAG:-det}
mkdir -p $OUT
: > $OUT/losses.txt
for i in $(seq 1 ${N:-24}); do
  if [ "${RANKS:-1}" -gt 1 ]; then   # several ranks on this one GPU: the native engine's direct xGMI kernel carries the exchange (RCCL refuses a shared device)
    env DDL_RCCL_SKIP_COMM=1 HVD_ALLREDUCE_ALGO=twoshot DDL_COMM_STREAM_PRIORITY=low HVD_COLLECTIVE_TIMEOUT_S=60 $ENVS timeout -k 10 300 \
      python -u -m torch.distributed.run --nnodes=1 --nproc-per-node $RANKS --master-addr 127.0.0.1 --master-port $((29600 + i)) \
      bench.py --gpus $RANKS --model ${MODEL:-resnet101} --steps ${STEPS:-30} --warmup 6 $ARGS > $OUT/bench.log 2>&1 || { tail -30 $OUT/bench.log; exit 1; }
    grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/last.json
  else
    env $ENVS timeout -k 10 200 python -u bench.py --model ${MODEL:-resnet101} --steps ${STEPS:-30} --warmup 6 $ARGS > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
    tail -1 $OUT/bench.log > $OUT/last.json
  fi
  cat $OUT/last.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["final_loss"], d["ms_per_step"])' | tee -a $OUT/losses.txt
done
echo "== ${MODEL:-resnet101} $ENVS: distinct final losses"
cut -d" " -f1 $OUT/losses.txt | sort | uniq -c
