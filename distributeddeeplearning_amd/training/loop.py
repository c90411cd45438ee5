"""The training / validation loops of the reference's PyTorch flavour, MI355X-first.

Parity targets in ``HorovodPytorch/src/imagenet_pytorch_horovod.py``:
``train`` ``:204-221`` (per-100-iteration line ``" duration({})  loss:{} total-samples: {}"``),
``validate`` ``:224-239``, ``_log_summary`` ``:242-255`` (the epoch summary block, identical
text so logs diff against the reference's) and the data / optimizer set-up of ``main``
``:281-346``.

What differs:
* the loss value is fetched from the device only when a line is logged (the reference calls
  ``loss.item()`` on those iterations too, but also pays a host round trip per step for the
  H2D copy; with the on-GPU synthetic generator a step issues no host<->device traffic);
* validation averages the loss over ranks and reports top-1 / top-5 (quirk #6), and reads the
  TEST directory (quirk #3);
* a steady-state images/sec figure that excludes the first ``log_interval`` iterations is
  reported next to the reference-style whole-epoch number.
"""
from __future__ import annotations

import torch

from .. import torch as hvd
from ..data import DevicePrefetcher, DeviceSyntheticLoader, FakeData, get_sampler
from ..data import imagenet as _imagenet
from ..ops import functional as ops
from ..utils import trace
from ..utils.timer import Timer

__all__ = ["train_epoch", "validate", "log_summary", "to_device_batch", "build_data",
           "build_optimizer", "build_graph_step"]

_TRAIN_MSG = " duration({})  loss:{} total-samples: {}"
_VAL_MSG = "validation duration({})  loss:{} total-samples: {}"


def to_device_batch(data, target, device, dtype):
    """Host batch (NCHW fp32, reference :211) → device NHWC compute dtype; device batches pass
    through untouched."""
    if data.device != device:
        data = data.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
    if data.dtype != dtype and data.is_floating_point():
        data = data.to(dtype)
    if data.dim() == 4 and device.type == "cuda":
        data = data.contiguous(memory_format=torch.channels_last)
    return data, target


def build_graph_step(cfg, model, optimizer, criterion, device, world):
    """A :class:`~.graph.GraphedTrainStep` for ``train_epoch(graph_step=...)`` or None.

    ``cfg.hip_graph``: ``on`` | ``off`` | ``auto``.  Measured on MI355X (profiles/r1_notes.md): replay
    beats eager issue where the host is the bottleneck (batch 16-32 per GPU: +24-28 %) and loses
    3-6 % from batch 64 up, where the eager path's second stream is worth more than the saved
    launches - so ``auto`` means one GPU, batch <= 32."""
    mode = str(getattr(cfg, "hip_graph", "auto")).strip().lower()
    if device.type != "cuda" or mode in ("off", "0", "false", "no"):
        return None
    if mode not in ("on", "1", "true", "yes") and not (world == 1 and cfg.batch_size <= 32):
        return None
    from .graph import GraphedTrainStep

    return GraphedTrainStep(model, optimizer, criterion, warmup=3)


def train_epoch(loader, model, criterion, optimizer, epoch, logger, device, dtype,
                log_interval=100, metrics=None, callbacks=None, max_steps=0, graph_step=None):
    """One epoch of the reference hot loop (:210-221).  Returns a dict with the last logged
    loss, the number of samples this rank processed and steady-state timing.  ``graph_step``
    (:func:`build_graph_step`) replaces the four calls of the step by one graph replay."""
    t = Timer()
    t.start()
    logger.set_epoch(epoch)
    steady = Timer()
    steady_on = False
    samples = steady_samples = 0
    last_loss = None
    loss = None
    for i, (data, target) in enumerate(loader):
        if max_steps and i >= max_steps:
            break
        if callbacks is not None:
            callbacks.on_batch_begin(i)
        data, target = to_device_batch(data, target, device, dtype)
        if graph_step is not None:
            loss = graph_step(data, target)
        else:
            # roctx ranges (DDL_ROCTX=1) for rocprofv3 --marker-trace timelines; no-ops otherwise
            optimizer.zero_grad()
            with trace.range("fwd"):
                output = model(data)
                loss = criterion(output, target)
            with trace.range("bwd"):
                loss.backward()
            with trace.range("step"):
                optimizer.step()
        n = len(data)
        samples += n
        if steady_on:
            steady_samples += n
        if callbacks is not None:
            callbacks.on_batch_end(i)
        if i % log_interval == 0:
            last_loss = float(loss.detach())  # the only device sync of the interval
            logger.info(_TRAIN_MSG.format(t.elapsed, last_loss, i * n))
            if metrics is not None:
                extra = {}
                cm = getattr(optimizer, "comm_metrics", None)
                if cm is not None:  # exposed communication time and per-bucket device ms
                    m = cm()
                    if m["comm_exposed_ms"] is not None or m["bucket_ms"]:
                        extra = {"comm_exposed_ms": m["comm_exposed_ms"], "bucket_ms": m["bucket_ms"]}
                metrics.write(kind="train", epoch=epoch, step=i, loss=last_loss,
                              interval_s=t.elapsed, **extra)
            t.start()
            if i == 0:  # the device is idle here: start the warm-up-free clock
                steady.start()
                steady_on = True
    steady_s = 0.0
    if loss is not None:
        last_loss = float(loss.detach())  # drains the device queue: epoch time is honest
    if steady_on:
        steady.stop()
        steady_s = steady.elapsed
    return {"loss": last_loss, "samples": samples, "steady_samples": steady_samples,
            "steady_seconds": steady_s}


def len_or(loader, default=1 << 60):
    try:
        return len(loader)
    except TypeError:
        return default


@torch.no_grad()
def validate(loader, model, criterion, logger, device, dtype, log_interval=100, max_steps=0):
    """Evaluation pass: rank-averaged loss, top-1 and top-5 accuracy."""
    t = Timer()
    t.start()
    was_training = model.training
    model.eval()
    tot_loss = torch.zeros((), dtype=torch.float64, device=device)
    correct = torch.zeros(2, dtype=torch.float64, device=device)
    seen = 0
    for i, (data, target) in enumerate(loader):
        if max_steps and i >= max_steps:
            break
        data, target = to_device_batch(data, target, device, dtype)
        output = model(data)
        loss = criterion(output, target)
        n = len(data)
        tot_loss += loss.detach().double() * n
        correct += ops.topk_hits(output, target, (1, 5))   # K21 on the GPU, no host sync
        seen += n
        if i % log_interval == 0:
            logger.info(_VAL_MSG.format(t.elapsed, float(loss), i * n))
            t.start()
    if was_training:
        model.train()
    stats = torch.cat([tot_loss.view(1), correct, torch.tensor([float(seen)], dtype=torch.float64,
                                                              device=device)])
    if hvd.is_initialized() and hvd.size() > 1:
        stats = hvd.allreduce(stats, op=hvd.Sum, name="validation.stats")
    total = max(float(stats[3]), 1.0)
    return {"loss": float(stats[0]) / total, "top1": float(stats[1]) / total,
            "top5": float(stats[2]) / total, "samples": int(total)}


def log_summary(logger, data_length, duration, batch_size, distributed, fake, steady=None):
    """The reference's epoch summary block (:242-255), text-identical, plus steady state."""
    world = hvd.size() if (distributed and hvd.is_initialized()) else 1
    logger.info("Data length:      {}".format(data_length))
    logger.info("Total duration:   {:.3f}".format(duration))
    logger.info("Total images/sec: {:.3f}".format(data_length / duration))
    logger.info("Batch size:       (Per GPU {}: Total {})".format(batch_size, world * batch_size))
    logger.info("Distributed:      {}".format("True" if distributed else "False"))
    logger.info("Num GPUs:         {:.3f}".format(world))
    logger.info("Dataset:          {}".format("Synthetic" if fake else "Imagenet"))
    if steady and steady.get("steady_seconds", 0) > 0:
        logger.info("Steady images/sec: {:.3f}".format(
            steady["steady_samples"] * world / steady["steady_seconds"]))


def build_data(cfg, device, dtype, logger, world, rank):
    """Training (and optional validation) loaders for a ``TrainConfig``.

    FAKE + GPU → the on-device Philox pool (K19); FAKE on CPU or ``DEVICE_DATA=False`` → the
    reference-semantics host ``FakeData`` through a DataLoader; otherwise ImageFolder loaders
    with the reference's transforms (:284-316).  Host loaders feeding a GPU are wrapped in a
    :class:`DevicePrefetcher` (upload + layout/dtype conversion of the next batches on a copy
    stream); real images then travel as uint8 and are normalised on the GPU."""
    on_gpu = device.type == "cuda"
    depth = getattr(cfg, "prefetch_depth", 2)

    def wrap(loader):
        if loader is None or not on_gpu or depth <= 0:
            return loader
        return DevicePrefetcher(loader, device, dtype, depth=depth)

    if cfg.fake:
        logger.info("Setting up fake loaders")
        if cfg.device_data and device.type == "cuda":
            train = DeviceSyntheticLoader(cfg.batch_size, length=cfg.fake_data_length,
                                          height=cfg.height, width=cfg.width,
                                          channels=cfg.channels, n_classes=cfg.num_classes,
                                          device=device, dtype=dtype, seed=cfg.seed, rank=rank,
                                          world=world, max_steps=cfg.max_steps)
            return train, None, cfg.fake_data_length, None
        ds = FakeData(n_classes=cfg.num_classes, dim=(cfg.height, cfg.width),
                      n_channels=cfg.channels, length=cfg.fake_data_length, seed=cfg.seed)
        sampler = get_sampler(ds, cfg.distributed, seed=cfg.seed)
        workers = 0 if device.type == "cpu" else cfg.num_workers
        return wrap(_imagenet.make_loader(ds, cfg.batch_size, sampler, workers)), None, len(ds), sampler
    logger.info("Setting up loaders")
    if not cfg.train_dir:
        raise ValueError("FAKE is False (the reference's default, imagenet_pytorch_horovod.py:49) and no "
                         "training directory is set: export AZ_BATCHAI_INPUT_TRAIN / INPUT_TRAIN_DIR, or "
                         "FAKE=True for synthetic data")
    raw = bool(on_gpu and depth > 0 and getattr(cfg, "uint8_transport", True))
    train_ds = _imagenet.ImageFolder(cfg.train_dir, _imagenet.train_transform(cfg.width, raw_uint8=raw))
    sampler = get_sampler(train_ds, cfg.distributed, seed=cfg.seed)
    train = wrap(_imagenet.make_loader(train_ds, cfg.batch_size, sampler, cfg.num_workers))
    val = None
    if cfg.test_dir:
        val_ds = _imagenet.ImageFolder(cfg.test_dir,
                                       _imagenet.val_transform(256, cfg.width, raw_uint8=raw))
        val = wrap(_imagenet.make_loader(val_ds, cfg.batch_size,
                                         get_sampler(val_ds, cfg.distributed, seed=cfg.seed),
                                         cfg.num_workers))
    return train, val, len(train_ds), sampler


def build_optimizer(cfg, model, world, fused=True):
    """SGD(lr × world, momentum 0.9) as in the reference (:333), fused over flat arenas."""
    from ..optim import FusedLARS, FusedSGD

    lr = cfg.lr * world
    if cfg.optimizer == "lars":
        return FusedLARS(model.named_parameters(), lr=lr, momentum=cfg.momentum,
                         weight_decay=cfg.weight_decay)
    if fused:
        return FusedSGD(model.named_parameters(), lr=lr, momentum=cfg.momentum,
                        weight_decay=cfg.weight_decay)
    return torch.optim.SGD(model.parameters(), lr=lr, momentum=cfg.momentum,
                           weight_decay=cfg.weight_decay)


def default_criterion(output, target):
    return ops.softmax_cross_entropy(output, target)
