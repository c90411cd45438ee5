# Training image of the Estimator-flavoured script (reference: HorovodTF/Docker/Dockerfile, a CUDA + TensorFlow +
# Horovod image whose entry point is imagenet_estimator_tf_horovod.py).  A thin layer over the framework's base
# image (deploy/docker/Dockerfile): model_fn / TRAIN-EVAL-PREDICT modes / hooks of training/estimator.py with the
# reference's model directory convention (rank 0 writes /model, other ranks a scratch directory:
# imagenet_estimator_tf_horovod.py:346-356).
ARG BASE_IMAGE=local/ddl-amd-training
FROM ${BASE_IMAGE}
ENV DDL_FLAVOUR=estimator \
    DISTRIBUTED=True \
    OUTPUT_MODEL_DIR=/model \
    VALIDATION=True
RUN mkdir -p /model
CMD ["bash", "-c", "ddlrun -np ${NP:-8} -H localhost:${NP:-8} python -u examples/imagenet_estimator.py"]
