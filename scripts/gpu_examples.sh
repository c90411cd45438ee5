#!/bin/bash
# The three reference-flavoured example scripts end to end on the GPU with synthetic data (short epochs).
set -o pipefail
mkdir -p gpurun_out/examples
export FAKE=True FAKE_DATA_LENGTH=4096 EPOCHS=2 BATCHSIZE=128
# one model directory per example: a shared one makes the second script resume from the first
# script's last checkpoint (epoch 2 of 2) and train nothing
rm -rf /tmp/ddl_examples_model
export OUTPUT_MODEL_DIR=/tmp/ddl_examples_model/pytorch
echo "== imagenet_pytorch.py (C1)"
timeout -k 10 300 python -u examples/imagenet_pytorch.py > gpurun_out/examples/pytorch.log 2>&1 || { tail -30 gpurun_out/examples/pytorch.log; exit 1; }
grep -i "images/sec\|Total duration\|checkpoint" gpurun_out/examples/pytorch.log | tail -6
echo "== imagenet_trainer_callbacks.py (C2, LARS + warm-up)"
export OUTPUT_MODEL_DIR=/tmp/ddl_examples_model/trainer
OPTIMIZER=lars timeout -k 10 300 python -u examples/imagenet_trainer_callbacks.py > gpurun_out/examples/trainer.log 2>&1 || { tail -30 gpurun_out/examples/trainer.log; exit 1; }
tail -5 gpurun_out/examples/trainer.log
echo "== imagenet_estimator.py (C3)"
export OUTPUT_MODEL_DIR=/tmp/ddl_examples_model/estimator
timeout -k 10 300 python -u examples/imagenet_estimator.py > gpurun_out/examples/estimator.log 2>&1 || { tail -30 gpurun_out/examples/estimator.log; exit 1; }
tail -5 gpurun_out/examples/estimator.log
ls -R /tmp/ddl_examples_model | head -20
