"""On-GPU synthetic ImageNet-shaped loader (kernel K19 of SURVEY §2.6).

The reference feeds synthetic data through a host pipeline: 5 forked DataLoader workers
collate 38.5 MB fp32 NCHW batches into ``/dev/shm``, a pin thread copies them to pinned memory
and ``data.cuda(non_blocking=True)`` ships them over PCIe
(``imagenet_pytorch_horovod.py:211,313-316``).  At MI355X step times (tens of
milliseconds for 256 images) that pipeline would be the bottleneck, so here the pool lives in
HBM: ``pool_size`` images are generated once **on the device** by a Philox HIP kernel directly
in the layout and dtype the stem convolution consumes (NHWC bf16), and a batch is a row gather
from the pool by the reference's random ``translation_index`` — same statistical semantics as
``FakeData`` (small pool, random index map of length ``FAKE_DATA_LENGTH``, rank-sharded), no
host memory traffic, no worker processes.

Batches are returned with logical shape ``[B, C, H, W]`` in ``channels_last`` memory format so
they are valid inputs for any PyTorch module as well as for the NHWC HIP kernels.
"""
from __future__ import annotations

import math

import torch

from ..utils.config import IMAGENET_TRAIN_LEN

__all__ = ["DeviceSyntheticLoader", "generate_pool"]


def generate_pool(pool_size, height, width, channels, n_classes, device, dtype, seed=42):
    """Uniform-[0,1) NHWC image pool + labels, generated on ``device``."""
    device = torch.device(device)
    if device.type == "cuda":
        from ..ops import kernels

        if kernels.enabled():
            images = torch.empty((pool_size, height, width, channels), dtype=dtype,
                                 device=device)
            labels = torch.empty((pool_size,), dtype=torch.int64, device=device)
            kernels.philox_uniform_(images, seed)
            kernels.philox_labels_(labels, n_classes, seed + 1)
            return images, labels
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    images = torch.rand((pool_size, height, width, channels), generator=g, device=device,
                        dtype=torch.float32).to(dtype)
    labels = torch.randint(0, n_classes, (pool_size,), generator=g, device=device,
                           dtype=torch.int64)
    return images, labels


class DeviceSyntheticLoader:
    """Iterable yielding ``(images, labels)`` device batches for one epoch.

    ``length`` is the *global* epoch length (reference default 1 281 167); with ``world`` ranks
    each rank iterates ``ceil(ceil(length/world)/batch_size)`` batches, the same count a
    ``DistributedSampler``-fed DataLoader would produce.
    """

    def __init__(self, batch_size, length=IMAGENET_TRAIN_LEN, pool_size=640, height=224,
                 width=224, channels=3, n_classes=1000, device="cuda", dtype=torch.bfloat16,
                 seed=42, rank=0, world=1, max_steps=0):
        self.batch_size = int(batch_size)
        self.length = int(length)
        self.rank, self.world = rank, world
        self.device = torch.device(device)
        self.n_classes = n_classes
        self.seed = seed
        self.epoch = 0
        self.per_rank = math.ceil(self.length / world)
        self.steps = math.ceil(self.per_rank / self.batch_size)
        if max_steps:
            self.steps = min(self.steps, int(max_steps))
        self.pool, self.pool_labels = generate_pool(pool_size, height, width, channels,
                                                    n_classes, self.device, dtype, seed)
        self._gen = torch.Generator(device=self.device)
        # HIP path: the index draw and both gathers are ONE in-tree kernel (ops/kernels.py::pool_gather);
        # the batch counter lives on the device, the stream key (seed, epoch, rank) travels by value
        self._hip = False
        if self.device.type == "cuda":
            from ..ops import kernels

            self._hip = kernels.enabled() and (self.pool[0].numel() * self.pool.element_size()) % 16 == 0 \
                and self.batch_size <= 65535
        if self._hip:
            self._state = torch.zeros(2, dtype=torch.int64, device=self.device)
            self._key = self._stream_key()

    def _stream_key(self):
        return self.seed * 1000003 + self.epoch * 8191 + self.rank

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.steps

    @property
    def samples_per_epoch(self):
        return min(self.steps * self.batch_size, self.per_rank)

    def batch(self, index=None):
        """One batch: a device-side gather from the pool (no host sync)."""
        if self._hip:
            from ..ops import kernels

            images = torch.empty((self.batch_size,) + tuple(self.pool.shape[1:]), dtype=self.pool.dtype,
                                 device=self.device)
            labels = torch.empty((self.batch_size,), dtype=torch.int64, device=self.device)
            kernels.pool_gather(self.pool, self.pool_labels, images, labels, self._state, self.batch_size, self._key)
            return images.permute(0, 3, 1, 2), labels
        idx = torch.randint(0, self.pool.shape[0], (self.batch_size,), generator=self._gen,
                            device=self.device)
        images = self.pool.index_select(0, idx).permute(0, 3, 1, 2)  # NHWC storage, NCHW view
        labels = self.pool_labels.index_select(0, idx)
        return images, labels

    def __iter__(self):
        self._gen.manual_seed(self.seed * 1000003 + self.epoch * 8191 + self.rank)
        if self._hip:   # a new epoch is a new Philox stream, counted from its first batch
            self._key = self._stream_key()
            self._state.zero_()
        for i in range(self.steps):
            yield self.batch(i)
