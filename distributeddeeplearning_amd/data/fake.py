"""Host-side synthetic ImageNet-shaped dataset with the reference's semantics.

Parity target: ``FakeData`` / ``_create_data`` / ``_create_labels``
(``HorovodPytorch/src/imagenet_pytorch_horovod.py:135-191``): a small pool of
``batch_size × num_batches`` (= 640 by default) uniform-[0,1) float32 NCHW images seeded with
42, ``n_classes`` random labels, and a random ``translation_index`` of length ``length``
(default 1 281 167) mapping epoch positions onto the pool, so that ``len(dataset)`` looks like
ImageNet while only 385 MB of host memory is used.  Every rank builds the identical pool; the
sampler shards indices.

Also here: ``FakeBatchIterator`` — the batch-level NHWC/one-hot variant of the Keras flavour
(``HorovodKeras/src/data_generator.py:22-52``) and the pre-batched NCHW generator of the TF
flavour (``imagenet_estimator_tf_horovod.py:295-345``).

This host path exists for API parity and CPU tests; the MI355X hot loop uses
``data.synthetic.DeviceSyntheticLoader`` which never touches host memory.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils.config import IMAGENET_TRAIN_LEN

__all__ = ["create_data", "create_labels", "FakeData", "FakeBatchIterator", "PrebatchedFakeStream"]


def create_data(batch_size, num_batches, dim, channels, rng, channels_last=False):
    n = batch_size * num_batches
    shape = (n, dim[0], dim[1], channels) if channels_last else (n, channels, dim[0], dim[1])
    return rng.random(shape, dtype=np.float32)


def create_labels(batch_size, num_batches, n_classes, rng):
    return rng.integers(0, n_classes, size=batch_size * num_batches, dtype=np.int64)


class FakeData(Dataset):
    """Map-style dataset: ``dataset[i] -> (image float32 [C,H,W], label int64)``."""

    def __init__(self, batch_size=32, num_batches=20, dim=(224, 224), n_channels=3,
                 n_classes=10, length=IMAGENET_TRAIN_LEN, seed=42, data_transform=None,
                 logger=None):
        self.dim = tuple(dim)
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.num_batches = num_batches
        rng = np.random.default_rng(seed)
        self._data = create_data(batch_size, num_batches, self.dim, n_channels, rng)
        self._labels = create_labels(batch_size, num_batches, n_classes, rng)
        self.translation_index = rng.integers(0, len(self._labels), size=length, dtype=np.int64)
        self._length = int(length)
        self._data_transform = data_transform
        if logger is not None:
            logger.info("Creating fake data {} labels and {} images".format(
                n_classes, len(self._data)))

    @property
    def pool_size(self):
        return len(self._labels)

    def __getitem__(self, idx):
        j = self.translation_index[idx]
        sample = self._data[j]
        if self._data_transform is not None:
            sample = self._data_transform(sample)
        else:
            sample = torch.from_numpy(sample)
        return sample, int(self._labels[j])

    def __len__(self):
        return self._length


class FakeBatchIterator:
    """Batch-level synthetic iterator (Keras ``FakeDataGenerator`` semantics).

    Yields ``(images, labels)`` batches from a pool of ``batch_size × num_batches`` images;
    ``channels_last=True`` gives NHWC, ``one_hot=True`` gives ``to_categorical`` labels;
    ``len(it)`` is the number of batches per epoch for ``length`` samples.
    """

    def __init__(self, batch_size=64, num_batches=20, dim=(224, 224), n_channels=3,
                 n_classes=1000, length=1000, shuffle=True, seed=42, channels_last=True,
                 one_hot=True):
        self.batch_size = batch_size
        self.n_classes = n_classes
        self.length = int(length)
        self.shuffle = shuffle
        self.one_hot = one_hot
        self._rng = np.random.default_rng(seed)
        self._data = create_data(batch_size, num_batches, dim, n_channels, self._rng,
                                 channels_last=channels_last)
        self._labels = create_labels(batch_size, num_batches, n_classes, self._rng)
        self._translation = self._rng.integers(0, len(self._labels), size=self.length)
        self._order = np.arange(self.length)
        self._cursor = 0
        self.on_epoch_end()

    def __len__(self):
        return int(np.ceil(self.length / float(self.batch_size)))

    def on_epoch_end(self):
        self._cursor = 0
        if self.shuffle:
            self._rng.shuffle(self._order)

    def __iter__(self):
        return self

    def __next__(self):
        if self._cursor >= len(self):
            self.on_epoch_end()
            raise StopIteration
        return self[self._cursor_advance()]

    def _cursor_advance(self):
        c = self._cursor
        self._cursor += 1
        return c

    def __getitem__(self, batch_index):
        sel = self._order[batch_index * self.batch_size: (batch_index + 1) * self.batch_size]
        pool_idx = self._translation[sel]
        x = torch.from_numpy(self._data[pool_idx])
        y = torch.from_numpy(self._labels[pool_idx])
        if self.one_hot:
            y = torch.nn.functional.one_hot(y, self.n_classes).to(torch.float32)
        return x, y


class PrebatchedFakeStream:
    """The TF flavour's synthetic input (``imagenet_estimator_tf_horovod.py:295-345``): a
    generator of ``num_batches`` (40) PRE-BATCHED ``[batch, 3, H, W]`` NCHW float32 arrays with
    int labels, then ``shuffle(buffer)`` — of whole batches —, ``repeat()`` and ``prefetch(n)``.

    ``length`` samples make an epoch (``len(stream)`` batches); iteration never ends when
    ``repeat=True`` (tf.data ``repeat()``), else stops after one epoch.  ``prefetch`` batches are
    produced ahead of the consumer by a background thread (the ``prefetch(_BUFFER)`` stage)."""

    def __init__(self, batch_size=64, num_batches=40, dim=(224, 224), n_channels=3, n_classes=1000,
                 length=None, shuffle_buffer=None, repeat=True, prefetch=2, seed=42):
        self.batch_size, self.num_batches = int(batch_size), int(num_batches)
        self.length = int(length if length is not None else batch_size * num_batches)
        self.repeat = bool(repeat)
        self.prefetch = max(int(prefetch), 0)
        rng = np.random.default_rng(seed)
        data = create_data(batch_size, num_batches, dim, n_channels, rng)
        labels = create_labels(batch_size, num_batches, n_classes, rng)
        shape = (self.num_batches, self.batch_size)
        self._images = data.reshape(shape + data.shape[1:])
        self._labels = labels.reshape(shape)
        # tf.data shuffle(buffer_size): a buffer of that many ELEMENTS (batches here); the
        # reference passes 40*batch, i.e. a buffer at least as large as the 40-element dataset
        buf = self.num_batches if shuffle_buffer is None else int(shuffle_buffer)
        self._buffer = max(min(buf, self.num_batches), 1)
        self._rng = np.random.default_rng(seed + 1)

    def __len__(self):
        return int(np.ceil(self.length / float(self.batch_size)))

    def _epoch_order(self):
        """One pass over the 40 batches through a shuffle buffer of ``_buffer`` elements."""
        pending = list(range(self.num_batches))
        buf, out = [], []
        while pending or buf:
            while pending and len(buf) < self._buffer:
                buf.append(pending.pop(0))
            k = int(self._rng.integers(0, len(buf)))
            out.append(buf.pop(k))
        return out

    def _generate(self):
        produced = 0
        per_epoch = len(self)
        while True:
            for b in self._epoch_order():
                yield torch.from_numpy(self._images[b]), torch.from_numpy(self._labels[b])
                produced += 1
                if not self.repeat and produced >= per_epoch:
                    return

    def __iter__(self):
        gen = self._generate()
        if self.prefetch == 0:
            return gen
        import queue
        import threading

        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        done = object()

        def worker():
            try:
                for item in gen:
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
            finally:
                if not stop.is_set():
                    q.put(done)

        th = threading.Thread(target=worker, daemon=True)
        th.start()

        def consume():
            try:
                while True:
                    item = q.get()
                    if item is done:
                        return
                    yield item
            finally:
                stop.set()

        return consume()

