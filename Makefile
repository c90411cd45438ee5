define PROJECT_HELP_MSG
Usage:
    make help        show this message
    make native      build the HIP kernels and the C++ engine in-tree (gfx950)
    make test        CPU test suite (pytest -m "not gpu")
    make gpu-test    GPU test suite on an MI355X (pytest -m gpu)
    make bench       flagship benchmark on the local GPU(s): make bench GPUS=8
    make smoke       2-rank local smoke run of the training script on synthetic data
    make build       build the training docker image       (image=..., see deploy/build.mk)
    make push        push the training docker image
    make flavours    build the per-script training images on top of it (trainer = Keras-flavoured script,
                     estimator = TF-Estimator-flavoured script; the base image runs the PyTorch-flavoured one)
    make control     build the control-plane image (docker CLI + Jupyter :9999)
    make jupyter     run Jupyter inside the control image
endef
export PROJECT_HELP_MSG
PWD:=$(shell pwd)
dockerhub?=local
data?=$(PWD)/data
GPUS?=1
image?=$(dockerhub)/ddl-amd-training
dockerpath?=.
dockerfile?=deploy/docker/Dockerfile
control_image:=$(dockerhub)/ddl-amd-control

help:
	@echo "$$PROJECT_HELP_MSG"

native:
	python build.py

test: native
	python -m pytest tests -x -q -m "not gpu"

gpu-test: native
	python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread

bench: native
ifeq ($(GPUS),1)
	python bench.py --gpus 1
else
	python -m torch.distributed.run --nnodes=1 --nproc-per-node $(GPUS) --master-addr 127.0.0.1 bench.py --gpus $(GPUS)
endif

smoke: native
	python -m distributeddeeplearning_amd.launch.run -np 2 -H localhost:2 -x DISTRIBUTED=True -x FAKE=True \
	    -x FAKE_DATA_LENGTH=2048 -x EPOCHS=1 python -u examples/imagenet_pytorch.py

include deploy/build.mk

flavours: build
	docker build -t $(image)-trainer --build-arg BASE_IMAGE=$(image) -f deploy/docker/trainer.Dockerfile .
	docker build -t $(image)-estimator --build-arg BASE_IMAGE=$(image) -f deploy/docker/estimator.Dockerfile .

control:
	docker build -t $(control_image) -f deploy/docker/control.Dockerfile .

jupyter:
	docker run -p 9999:9999 -e EXT_PWD=$(PWD) -e EXT_DATA=$(data) -e DOCKER_REPOSITORY=$(dockerhub) \
	    -v $(PWD):/workspace -v $(data):/data -v /var/run/docker.sock:/var/run/docker.sock \
	    -it $(control_image) bash -c "jupyter notebook"

.PHONY: help native test gpu-test bench smoke flavours control jupyter
