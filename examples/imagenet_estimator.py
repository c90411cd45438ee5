#!/usr/bin/env python
"""The reference's TF-Estimator flavour (``HorovodTF/src/imagenet_estimator_tf_horovod.py``)
on this framework: ``model_fn`` with TRAIN / EVAL / PREDICT modes (:170-220), ResNet v1 built
by depth (:163-167), momentum optimizer scaled by world size and wrapped for gradient
averaging (:149-160), ``BroadcastGlobalVariablesHook(0)`` (:377-384), rank-0 ``model_dir``.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import distributeddeeplearning_amd.torch as hvd  # noqa: E402
from distributeddeeplearning_amd.models import resnet  # noqa: E402
from distributeddeeplearning_amd.ops import functional as ops  # noqa: E402
from distributeddeeplearning_amd.optim import FusedSGD  # noqa: E402
from distributeddeeplearning_amd.training import (BroadcastGlobalVariablesHook, Estimator,  # noqa: E402
                                                  EstimatorSpec, ModeKeys, build_data)
from distributeddeeplearning_amd.training.loop import log_summary, to_device_batch  # noqa: E402
from distributeddeeplearning_amd.utils import Timer, TrainConfig, get_logger  # noqa: E402
from distributeddeeplearning_amd.utils import checkpoint  # noqa: E402


def model_fn(features, labels, mode, params):
    st = params["state"]
    if "model" not in st:  # first call: create the variables
        model = resnet(params.get("depth", 50), num_classes=params["classes"])
        dev, dt = params["device"], params["dtype"]
        model.to_compute(dev, dt) if dt == torch.bfloat16 else model.to(dev)
        opt = FusedSGD(model.named_parameters(), lr=params["learning_rate"] * hvd.size(),
                       momentum=params["momentum"])
        st["model"] = model
        st["optimizer"] = hvd.DistributedOptimizer(opt, named_parameters=model.named_parameters())
    model, opt = st["model"], st["optimizer"]
    features, labels = to_device_batch(features, labels if labels is not None else torch.zeros(0),
                                       params["device"], params["dtype"])
    if mode == ModeKeys.PREDICT:
        model.eval()
        logits = model(features).float()
        return EstimatorSpec(mode, predictions={"classes": logits.argmax(1),
                                                "probabilities": logits.softmax(1)})
    model.train(mode == ModeKeys.TRAIN)
    logits = model(features)
    loss = ops.softmax_cross_entropy(logits, labels)
    if mode == ModeKeys.EVAL:
        acc = (logits.float().argmax(1) == labels).float().mean()
        return EstimatorSpec(mode, loss=loss, eval_metric_ops={"accuracy": acc})

    def train_op():
        opt.zero_grad()
        loss.backward()
        opt.step()

    return EstimatorSpec(mode, loss=loss, train_op=train_op)


def main(argv=None):
    cfg = TrainConfig.from_args(argv)
    logger = get_logger()
    hvd.init()
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    dtype = torch.bfloat16 if (cfg.dtype == "bf16" and use_cuda) else torch.float32
    train_loader, val_loader, data_length, _ = build_data(cfg, device, dtype, logger, hvd.size(), hvd.rank())
    params = {"classes": cfg.num_classes, "learning_rate": cfg.lr, "momentum": cfg.momentum,
              "device": device, "dtype": dtype, "depth": int(cfg.model.replace("resnet", "") or 50)}
    est = Estimator(model_fn, model_dir=checkpoint.model_dir(cfg.output_dir, cfg.temp_dir), params=params)
    steps = cfg.epochs * data_length // (cfg.batch_size * hvd.size())
    if cfg.max_steps:
        steps = min(steps, cfg.max_steps)
    hooks = [BroadcastGlobalVariablesHook(0)] if hvd.size() > 1 else []
    logger.info("Training ...")
    with Timer(output=logger.info, prefix="Training") as t:
        est.train(lambda: iter(train_loader), steps=steps, hooks=hooks)
        if use_cuda:
            torch.cuda.synchronize()
    log_summary(logger, steps * cfg.batch_size * hvd.size(), t.elapsed, cfg.batch_size,
                hvd.size() > 1, cfg.fake)
    if cfg.validation and val_loader is not None and hvd.rank() == 0:
        logger.info("Validation: {}".format(est.evaluate(lambda: iter(val_loader))))
    hvd.shutdown()


if __name__ == "__main__":
    main()
