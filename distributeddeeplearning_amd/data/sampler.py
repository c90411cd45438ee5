"""Index samplers: rank-sharded for distributed runs, random otherwise.

Parity target: ``_get_sampler`` (``imagenet_pytorch_horovod.py:258-264``) which returns
``DistributedSampler(dataset, num_replicas=hvd.size(), rank=hvd.rank())`` or a
``RandomSampler``, and ``train_sampler.set_epoch(epoch)`` each epoch (``:353-354``).  The
sharding is pure index arithmetic: a permutation seeded by ``seed + epoch``,
padded by wrap-around to a multiple of the world size, strided by rank — the same contract as
torch's DistributedSampler, so every rank sees ``ceil(N/size)`` samples and the union covers
the dataset.
"""
from __future__ import annotations

import math

import torch
from torch.utils.data import Sampler

__all__ = ["ShardedSampler", "EpochRandomSampler", "get_sampler"]


class ShardedSampler(Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True, seed=0,
                 drop_last=False):
        if num_replicas is None or rank is None:
            from ..parallel import backend as _b

            num_replicas = _b.size() if num_replicas is None else num_replicas
            rank = _b.rank() if rank is None else rank
        if not 0 <= rank < num_replicas:
            raise ValueError("rank %d outside [0, %d)" % (rank, num_replicas))
        self.n = len(dataset)
        self.num_replicas = num_replicas
        self.rank = rank
        self.shuffle = shuffle
        self.seed = seed
        self.drop_last = drop_last
        self.epoch = 0
        if drop_last and self.n % num_replicas:
            self.num_samples = self.n // num_replicas
        else:
            self.num_samples = math.ceil(self.n / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _global_order(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g)
        else:
            order = torch.arange(self.n)
        if self.drop_last:
            return order[: self.total_size]
        pad = self.total_size - self.n
        if pad > 0:
            reps = math.ceil(pad / max(self.n, 1))
            order = torch.cat([order, order.repeat(reps)[:pad]])
        return order

    def __iter__(self):
        return iter(self._global_order()[self.rank: self.total_size: self.num_replicas].tolist())

    def __len__(self):
        return self.num_samples


class EpochRandomSampler(Sampler):
    """Single-process random permutation, reshuffled per epoch."""

    def __init__(self, dataset, seed=0):
        self.n = len(dataset)
        self.seed = seed
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        return iter(torch.randperm(self.n, generator=g).tolist())

    def __len__(self):
        return self.n


def get_sampler(dataset, is_distributed, seed=0):
    if is_distributed:
        return ShardedSampler(dataset, seed=seed)
    return EpochRandomSampler(dataset, seed=seed)
