# Training image of the Keras-flavoured script (reference: HorovodKeras/Docker/Dockerfile, a CUDA + TensorFlow +
# Keras + Horovod image whose entry point is imagenet_keras_horovod.py).  Here the flavour is a thin layer over the
# framework's base image (deploy/docker/Dockerfile: ROCm + PyTorch-ROCm + the in-tree HIP kernels and C++ engine):
# the Trainer/callbacks script with its defaults - L2 weight decay, 5 warm-up epochs, step schedule, rank-0
# checkpoints under /checkpoints, metric averaging (reference imagenet_keras_horovod.py:208-224, 287-324).
ARG BASE_IMAGE=local/ddl-amd-training
FROM ${BASE_IMAGE}
ENV DDL_FLAVOUR=trainer \
    DISTRIBUTED=True \
    WARMUP_EPOCHS=5 \
    OUTPUT_MODEL_DIR=/checkpoints \
    VALIDATION=True
RUN mkdir -p /checkpoints
# one rank per GPU of the node; multi-node jobs pass -H / --hostfile to ddlrun instead (tools/submit_job.py)
CMD ["bash", "-c", "ddlrun -np ${NP:-8} -H localhost:${NP:-8} python -u examples/imagenet_trainer_callbacks.py"]
