#!/usr/bin/env python
"""ResNet-50 synchronous data-parallel training on ImageNet or synthetic data — the
behaviour-equivalent of the reference's ``HorovodPytorch/src/imagenet_pytorch_horovod.py``.

Same environment contract (``DISTRIBUTED``, ``FAKE``, ``FAKE_DATA_LENGTH``, ``EPOCHS``,
``AZ_BATCHAI_INPUT_TRAIN/TEST``, ``AZ_BATCHAI_OUTPUT_MODEL`` ...), same log
lines and summary block, same hot loop; every flag is also available as ``--flag``.

    # single GPU, synthetic data
    FAKE=True python examples/imagenet_pytorch.py
    # 8 GPUs on one node (the reference: mpirun -np 8 -H localhost:8 ...)
    python -m distributeddeeplearning_amd.launch.run -np 8 -H localhost:8 \
        -x DISTRIBUTED=True -x FAKE=True python -u examples/imagenet_pytorch.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import distributeddeeplearning_amd.torch as hvd  # noqa: E402
from distributeddeeplearning_amd.models import build_model  # noqa: E402
from distributeddeeplearning_amd.training import (build_data, build_graph_step, build_optimizer,  # noqa: E402
                                                  log_summary, train_epoch, validate)
from distributeddeeplearning_amd.training.loop import default_criterion  # noqa: E402
from distributeddeeplearning_amd.utils import MetricsWriter, Timer, TrainConfig, get_logger  # noqa: E402
from distributeddeeplearning_amd.utils import checkpoint  # noqa: E402


def main(argv=None):
    cfg = TrainConfig.from_args(argv)
    logger = get_logger()
    use_cuda = torch.cuda.is_available()
    if cfg.distributed:
        hvd.init()
        logger.info("Runnin Distributed")
        if use_cuda:
            torch.cuda.set_device(hvd.local_rank() % torch.cuda.device_count())
    else:
        hvd.init()  # single-process: collectives are no-ops
    torch.manual_seed(cfg.seed)
    world, rank = hvd.size(), hvd.rank()
    device = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    dtype = torch.bfloat16 if (cfg.dtype == "bf16" and use_cuda) else torch.float32
    logger.info("PyTorch version {}".format(torch.__version__))

    train_loader, val_loader, data_length, sampler = build_data(cfg, device, dtype, logger, world, rank)

    logger.info("Loading model")
    model = build_model(cfg.model, num_classes=cfg.num_classes)
    if dtype == torch.bfloat16:
        model.to_compute(device, dtype)
    else:
        model.to(device)

    if cfg.distributed:
        hvd.broadcast_parameters(model.state_dict(), root_rank=0)

    optimizer = build_optimizer(cfg, model, world)
    if cfg.distributed:
        optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters())

    out_dir = checkpoint.model_dir(cfg.output_dir, cfg.temp_dir)
    start_epoch = checkpoint.resume(cfg.output_dir, model, optimizer) if (cfg.resume and cfg.output_dir) else 0
    if start_epoch:
        logger.info("Resuming from epoch {}".format(start_epoch))
    metrics = MetricsWriter(cfg.metrics_path)
    criterion = default_criterion

    # HIP_GRAPH=auto: one GPU and a small batch -> the step is replayed as one HIP graph
    graph_step = build_graph_step(cfg, model, optimizer, criterion, device, world)
    logger.info("Training ...")
    for epoch in range(start_epoch, cfg.epochs):
        with Timer(output=logger.info, prefix="Training") as t:
            model.train()
            if sampler is not None and hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            if hasattr(train_loader, "set_epoch"):
                train_loader.set_epoch(epoch)
            stats = train_epoch(train_loader, model, criterion, optimizer, epoch, logger, device,
                                dtype, cfg.log_interval, metrics, max_steps=cfg.max_steps,
                                graph_step=graph_step)
        seen = stats["samples"] * world if cfg.max_steps else data_length
        log_summary(logger, seen, t.elapsed, cfg.batch_size, cfg.distributed, cfg.fake, stats)
        metrics.write(kind="epoch", epoch=epoch, seconds=t.elapsed, images_per_sec=seen / t.elapsed)
        if cfg.output_dir and rank == 0:
            checkpoint.save(out_dir, epoch + 1, model, optimizer)

    if val_loader is not None and (cfg.validation or not cfg.fake):
        res = validate(val_loader, model, criterion, logger, device, dtype, cfg.log_interval)
        logger.info("Validation loss: {loss:.4f} top-1: {top1:.4f} top-5: {top5:.4f}".format(**res))
    metrics.close()
    hvd.shutdown()


if __name__ == "__main__":
    main()
