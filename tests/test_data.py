"""L1 data: FakeData semantics, samplers, synthetic loader, image folder."""
import numpy as np
import pytest
import torch

from distributeddeeplearning_amd.data import (DeviceSyntheticLoader, FakeBatchIterator, FakeData,
                                              ShardedSampler, get_sampler)
from distributeddeeplearning_amd.data import imagenet


def test_fakedata_pool_and_index_map():
    ds = FakeData(n_classes=1000, length=5000, dim=(8, 8))
    assert len(ds) == 5000 and ds.pool_size == 32 * 20  # reference pool: batch 32 x 20 batches
    x, y = ds[17]
    assert x.shape == (3, 8, 8) and x.dtype == torch.float32 and 0 <= y < 1000
    assert float(x.min()) >= 0.0 and float(x.max()) < 1.0
    # deterministic (every rank builds the identical pool, reference :162-166)
    ds2 = FakeData(n_classes=1000, length=5000, dim=(8, 8))
    assert torch.equal(ds[123][0], ds2[123][0]) and ds[123][1] == ds2[123][1]
    # index map reuses pool entries
    j = ds.translation_index
    assert j.min() >= 0 and j.max() < ds.pool_size and len(np.unique(j)) <= ds.pool_size


def test_fakedata_default_length_is_imagenet():
    from distributeddeeplearning_amd.utils.config import IMAGENET_TRAIN_LEN

    assert IMAGENET_TRAIN_LEN == 1281167


def test_fake_batch_iterator_keras_semantics():
    it = FakeBatchIterator(batch_size=4, num_batches=3, dim=(8, 8), n_classes=10, length=10)
    assert len(it) == 3
    x, y = it[0]
    assert x.shape == (4, 8, 8, 3) and y.shape == (4, 10) and torch.all(y.sum(1) == 1)
    assert sum(1 for _ in it) == 3


def test_sharded_sampler_partitions_and_reshuffles():
    ds = list(range(103))
    shards = [list(ShardedSampler(ds, num_replicas=4, rank=r, seed=1)) for r in range(4)]
    assert all(len(s) == 26 for s in shards)  # ceil(103/4), padded like torch's sampler
    assert set(sum(shards, [])) == set(range(103))
    s = ShardedSampler(ds, num_replicas=4, rank=0, seed=1)
    a = list(s)
    s.set_epoch(1)
    assert list(s) != a
    with pytest.raises(ValueError):
        ShardedSampler(ds, num_replicas=2, rank=2)


def test_get_sampler_non_distributed_is_random_permutation():
    s = get_sampler(list(range(10)), False, seed=3)
    assert sorted(s) == list(range(10))


def test_device_synthetic_loader_cpu():
    ld = DeviceSyntheticLoader(4, length=40, pool_size=8, height=8, width=8, device="cpu",
                               dtype=torch.float32, rank=1, world=2)
    assert len(ld) == 5  # ceil(ceil(40/2)/4)
    batches = list(ld)
    assert len(batches) == 5
    x, y = batches[0]
    assert x.shape == (4, 3, 8, 8) and x.stride(1) == 1  # NCHW view of NHWC storage
    assert y.dtype == torch.int64 and int(y.max()) < 1000


def test_image_folder_and_transforms(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    for cls in ("n01", "n02"):
        (tmp_path / cls).mkdir()
        for i in range(3):
            PIL.fromarray((np.random.rand(40, 50, 3) * 255).astype("uint8")).save(
                tmp_path / cls / ("%d.png" % i))
    ds = imagenet.ImageFolder(str(tmp_path), imagenet.train_transform(16))
    assert len(ds) == 6 and ds.classes == ["n01", "n02"]
    x, y = ds[4]
    assert x.shape == (3, 16, 16) and y == 1
    v = imagenet.ImageFolder(str(tmp_path), imagenet.val_transform(20, 16))
    assert v[0][0].shape == (3, 16, 16)
    with pytest.raises(FileNotFoundError):
        imagenet.ImageFolder(str(tmp_path / "missing"))


def test_csv_dataset(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    PIL.fromarray(np.zeros((8, 8, 3), dtype="uint8")).save(tmp_path / "a.png")
    (tmp_path / "train.csv").write_text("filenames,num_id\na.png,7\n")
    ds = imagenet.CsvImageDataset(str(tmp_path / "train.csv"))
    assert len(ds) == 1 and ds[0][1] == 6  # labels are 1-based in the file (reference :98-132)


def test_prebatched_fake_stream_tf_semantics():
    """TF flavour (imagenet_estimator_tf_horovod.py:295-345): 40 pre-batched NCHW batches,
    batch-level shuffle, repeat, prefetch."""
    from distributeddeeplearning_amd.data import PrebatchedFakeStream

    s = PrebatchedFakeStream(batch_size=4, num_batches=10, dim=(8, 8), length=40, prefetch=2)
    assert len(s) == 10
    it = iter(s)
    seen = []
    for _ in range(25):  # repeat(): runs past one epoch
        x, y = next(it)
        assert x.shape == (4, 3, 8, 8) and x.dtype == torch.float32 and y.shape == (4,)
        assert 0.0 <= float(x.min()) and float(x.max()) < 1.0
        seen.append(float(x.sum()))
    it.close()
    # every epoch is a permutation of the same 10 batches
    assert sorted(seen[:10]) == sorted(seen[10:20])
    assert seen[:10] != seen[10:20]
    one = list(PrebatchedFakeStream(batch_size=4, num_batches=10, dim=(8, 8), length=22,
                                    repeat=False, prefetch=0))
    assert len(one) == 6  # ceil(22 / 4) batches, then stop
    small_buf = PrebatchedFakeStream(batch_size=2, num_batches=6, dim=(4, 4), shuffle_buffer=1,
                                     repeat=False, prefetch=0)
    sums = [float(x.sum()) for x, _ in small_buf]
    ref = [float(torch.from_numpy(small_buf._images[i]).sum()) for i in range(6)]
    assert sums == ref  # buffer of one element = no shuffling


def test_keras_transforms():
    """Keras flavour augmentation (imagenet_keras_horovod.py:122-149): shift/zoom/flip affine with
    nearest-edge fill and caffe-style preprocess_input."""
    import random

    from PIL import Image

    from distributeddeeplearning_amd.data.imagenet import keras_train_transform, keras_val_transform

    arr = (np.arange(64 * 48 * 3) % 251).astype(np.uint8).reshape(48, 64, 3)
    img = Image.fromarray(arr)
    t = keras_train_transform(size=32, rng=random.Random(3))(img)
    assert t.shape == (3, 32, 32) and t.dtype == torch.float32
    assert float(t.min()) >= -124.0 and float(t.max()) <= 255.0 - 103.0
    # identity parameters reproduce resize + preprocess_input exactly (BGR order, mean-subtracted)
    ident = keras_train_transform(size=32, width_shift=0.0, height_shift=0.0, zoom=0.0,
                                  horizontal_flip=False)(img)
    base = np.asarray(img.resize((32, 32), Image.NEAREST), dtype=np.float32)
    want = base[:, :, ::-1] - np.array([103.939, 116.779, 123.68], dtype=np.float32)
    assert np.allclose(ident.numpy(), want.transpose(2, 0, 1), atol=1e-3)
    v = keras_val_transform(size=32)(img)
    assert v.shape == (3, 32, 32)
    # zoom 0.875 about the centre keeps the centre pixel neighbourhood in place
    assert np.allclose(v.numpy()[:, 15:17, 15:17].mean(), ident.numpy()[:, 15:17, 15:17].mean(), atol=30)


def test_device_prefetcher_cpu_passthrough_and_uint8_transport(tmp_path):
    """data/prefetch.py on a CPU device: same batches in the same order, ``len`` and ``sampler``
    forwarded; uint8 batches (``raw_uint8`` transforms) come out normalised like the float path."""
    from PIL import Image
    from distributeddeeplearning_amd.data import DevicePrefetcher
    from distributeddeeplearning_amd.data import imagenet as im
    from distributeddeeplearning_amd.utils.config import RGB_MEAN, RGB_SD

    batches = [(torch.full((2, 3, 4, 4), float(i)), torch.tensor([i, i + 1])) for i in range(5)]
    pf = DevicePrefetcher(batches, "cpu", dtype=torch.float32)
    got = list(pf)
    assert len(pf) == 5 and len(got) == 5
    for (a, b), (c, d) in zip(batches, got):
        assert torch.equal(a, c) and torch.equal(b, d)

    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 256, size=(40, 48, 3), dtype=np.uint8))
    raw = im.val_transform(32, 24, raw_uint8=True)(img)
    ref = im.val_transform(32, 24)(img)
    assert raw.dtype == torch.uint8 and raw.shape == (3, 24, 24)
    out, _ = next(iter(DevicePrefetcher([(raw[None], torch.tensor([0]))], "cpu", dtype=torch.float32,
                                        mean=RGB_MEAN, sd=RGB_SD)))
    assert out.dtype == torch.float32 and torch.allclose(out[0], ref, atol=1e-6)


def test_build_data_wraps_host_loaders_only_on_gpu(tmp_path):
    """training.build_data: on a CPU device host loaders come back bare (no prefetcher, float
    batches); the prefetcher forwards ``len`` and ``sampler`` of a real DataLoader."""
    import logging
    from distributeddeeplearning_amd.data import DevicePrefetcher
    from distributeddeeplearning_amd.training import build_data
    from distributeddeeplearning_amd.utils import TrainConfig

    cfg = TrainConfig(fake=True, fake_data_length=16, batch_size=4, device_data=False, num_workers=0,
                      height=32, width=32)
    train, val, n, sampler = build_data(cfg, torch.device("cpu"), torch.float32,
                                        logging.getLogger("t"), 1, 0)
    assert not isinstance(train, DevicePrefetcher) and val is None and n == 16
    data, target = next(iter(train))
    assert data.dtype == torch.float32 and tuple(data.shape) == (4, 3, 32, 32)
    pf = DevicePrefetcher(train, "cpu", dtype=torch.float32)
    assert len(pf) == len(train) == 4 and pf.sampler is train.sampler
    assert sum(len(t) for _, t in pf) == 16
