"""Real-data path: class-per-directory image folders with the reference's transforms.

Parity target: ``datasets.ImageFolder`` + transforms at
``imagenet_pytorch_horovod.py:284-309``: train = RandomResizedCrop(224) +
RandomHorizontalFlip + ToTensor + Normalize(mean, sd); validation = Resize(256) +
CenterCrop(224) + ToTensor + Normalize.  torchvision is not available in this environment, so
the folder scan and the transforms are implemented here on PIL + torch.  The reference reads
its *validation* set from the TRAIN directory; here validation reads the
TEST directory.

Also: ``CsvImageDataset`` for the reference's (dead) ``train.csv``/``validation.csv`` loaders
(``:98-132``, C12: columns ``filenames,num_id``, 1-based labels shifted to 0-based).

The device side of this path is ``ops.nchw_to_nhwc_normalize`` (kernel K20): loaders may emit
raw uint8/float NCHW batches and the mean/sd normalisation + NHWC bf16 conversion happens in
one HIP kernel after the H2D copy.
"""
from __future__ import annotations

import csv
import math
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils.config import RGB_MEAN, RGB_SD

__all__ = ["ImageFolder", "CsvImageDataset", "train_transform", "val_transform", "keras_train_transform", "keras_val_transform",
           "make_loader", "IMG_EXTENSIONS"]

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def _pil():
    from PIL import Image

    return Image


def _to_tensor(img, normalize=True, raw_uint8=False):
    if raw_uint8:  # CHW uint8: scaling + normalisation happen on the GPU (kernel K20, data/prefetch.py)
        arr = np.asarray(img.convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    arr = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0
    t = torch.from_numpy(arr).permute(2, 0, 1).contiguous()
    if normalize:
        mean = torch.tensor(RGB_MEAN).view(3, 1, 1)
        sd = torch.tensor(RGB_SD).view(3, 1, 1)
        t = (t - mean) / sd
    return t


def _resize_shorter(img, size):
    w, h = img.size
    if w <= h:
        nw, nh = size, max(int(round(h * size / w)), 1)
    else:
        nw, nh = max(int(round(w * size / h)), 1), size
    return img.resize((nw, nh), _pil().BILINEAR)


def _center_crop(img, size):
    w, h = img.size
    left, top = max((w - size) // 2, 0), max((h - size) // 2, 0)
    return img.crop((left, top, left + size, top + size))


def train_transform(size=224, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), normalize=True,
                    rng=None, raw_uint8=False):
    """RandomResizedCrop(size) + RandomHorizontalFlip + ToTensor + Normalize.  ``raw_uint8``: stop
    after the flip and return the CHW uint8 pixels (a quarter of the bytes over PCIe; ToTensor and
    Normalize then run on the GPU, ``data.prefetch.DevicePrefetcher``)."""
    rng = rng or random

    def apply(img):
        Image = _pil()
        w, h = img.size
        area = w * h
        box = None
        for _ in range(10):
            target = area * rng.uniform(*scale)
            logr = (math.log(ratio[0]), math.log(ratio[1]))
            ar = math.exp(rng.uniform(*logr))
            cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                left, top = rng.randint(0, w - cw), rng.randint(0, h - ch)
                box = (left, top, left + cw, top + ch)
                break
        if box is None:  # fallback: central crop at the closest legal aspect ratio
            in_ratio = w / h
            if in_ratio < ratio[0]:
                cw, ch = w, int(round(w / ratio[0]))
            elif in_ratio > ratio[1]:
                cw, ch = int(round(h * ratio[1])), h
            else:
                cw, ch = w, h
            left, top = (w - cw) // 2, (h - ch) // 2
            box = (left, top, left + cw, top + ch)
        img = img.crop(box).resize((size, size), Image.BILINEAR)
        if rng.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        return _to_tensor(img, normalize, raw_uint8)

    return apply


_CAFFE_MEAN_BGR = (103.939, 116.779, 123.68)


def _caffe_preprocess(img):
    """``keras.applications.resnet50.preprocess_input``: RGB -> BGR, subtract the ImageNet channel
    means on the 0-255 scale, no scaling; returns a float32 CHW tensor."""
    import numpy as np

    a = np.asarray(img.convert("RGB"), dtype=np.float32)[:, :, ::-1] - np.asarray(_CAFFE_MEAN_BGR, np.float32)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


def keras_train_transform(size=224, width_shift=0.33, height_shift=0.33, zoom=0.5,
                          horizontal_flip=True, preprocess="caffe", rng=None):
    """The Keras flavour's training augmentation (``imagenet_keras_horovod.py:136-149``):
    ``ImageDataGenerator(width_shift_range=0.33, height_shift_range=0.33, zoom_range=0.5,
    horizontal_flip=True, preprocessing_function=resnet50.preprocess_input)`` on images resized
    to ``size`` x ``size`` (``flow_from_directory(target_size=...)``): a random affine map —
    zoom factors drawn independently per axis from [1-zoom, 1+zoom], shifts as fractions of the
    image size — about the image centre with nearest-edge fill, then a coin-flip mirror."""
    rng = rng or random

    def apply(img):
        Image = _pil()
        img = img.convert("RGB").resize((size, size), Image.NEAREST)
        zx, zy = rng.uniform(1 - zoom, 1 + zoom), rng.uniform(1 - zoom, 1 + zoom)
        tx, ty = rng.uniform(-width_shift, width_shift) * size, rng.uniform(-height_shift, height_shift) * size
        c = (size - 1) / 2.0
        # output pixel (x, y) samples input (zx*(x-c) + c + tx, zy*(y-c) + c + ty)
        coeffs = (zx, 0.0, c - zx * c + tx, 0.0, zy, c - zy * c + ty)
        img = _affine_nearest_fill(img, coeffs, size)
        if horizontal_flip and rng.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        return _caffe_preprocess(img) if preprocess == "caffe" else _to_tensor(img, True)

    return apply


def keras_val_transform(size=224, zoom=0.875, preprocess="caffe"):
    """Validation generator of the Keras flavour (``imagenet_keras_horovod.py:122-133``): fixed
    zoom 0.875 about the centre (a central 87.5 % crop scaled back up) + ``preprocess_input``."""

    def apply(img):
        Image = _pil()
        img = img.convert("RGB").resize((size, size), Image.NEAREST)
        c = (size - 1) / 2.0
        img = _affine_nearest_fill(img, (zoom, 0.0, c - zoom * c, 0.0, zoom, c - zoom * c), size)
        return _caffe_preprocess(img) if preprocess == "caffe" else _to_tensor(img, True)

    return apply


def _affine_nearest_fill(img, coeffs, size):
    """Affine resampling with Keras' default ``fill_mode='nearest'`` (coordinates clamp to the
    edge); PIL fills with a constant, so the image is edge-padded first and the map shifted."""
    import numpy as np

    Image = _pil()
    pad = size  # shifts of +-0.33 and zoom 1.5 stay inside one image width of padding
    a = np.asarray(img)
    a = np.pad(a, ((pad, pad), (pad, pad), (0, 0)), mode="edge")
    big = Image.fromarray(a)
    a0, a1, a2, b0, b1, b2 = coeffs
    shifted = (a0, a1, a2 + pad, b0, b1, b2 + pad)
    return big.transform((size, size), Image.AFFINE, shifted, resample=Image.BILINEAR)


def val_transform(resize=256, crop=224, normalize=True, raw_uint8=False):
    """Resize(256) + CenterCrop(224) + ToTensor + Normalize (``raw_uint8`` as in train_transform)."""

    def apply(img):
        return _to_tensor(_center_crop(_resize_shorter(img, resize), crop), normalize, raw_uint8)

    return apply


class ImageFolder(Dataset):
    """``root/<class_name>/<image>`` dataset; classes sorted by name → index."""

    def __init__(self, root, transform=None, extensions=IMG_EXTENSIONS):
        if not root or not os.path.isdir(root):
            raise FileNotFoundError("image folder %r does not exist" % (root,))
        self.root = root
        self.transform = transform
        self.classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not self.classes:
            raise FileNotFoundError("no class directories under %r" % root)
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            cdir = os.path.join(root, c)
            for dirpath, _, files in sorted(os.walk(cdir)):
                for f in sorted(files):
                    if f.lower().endswith(extensions):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))
        if not self.samples:
            raise FileNotFoundError("no images with extensions %s under %r" % (extensions, root))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        path, target = self.samples[idx]
        with open(path, "rb") as fh:
            img = _pil().open(fh)
            img.load()
        if self.transform is not None:
            img = self.transform(img)
        return img, target


class CsvImageDataset(Dataset):
    """``filenames,num_id`` CSV listing (labels are 1-based in the file)."""

    def __init__(self, csv_path, root=None, transform=None):
        self.root = root or os.path.dirname(os.path.abspath(csv_path))
        self.transform = transform
        self.samples = []
        with open(csv_path, newline="") as fh:
            for row in csv.DictReader(fh):
                self.samples.append((row["filenames"], int(row["num_id"]) - 1))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        rel, target = self.samples[idx]
        path = rel if os.path.isabs(rel) else os.path.join(self.root, rel)
        img = _pil().open(path)
        img.load()
        if self.transform is not None:
            img = self.transform(img)
        return img, target


def make_loader(dataset, batch_size, sampler, num_workers=5, pin_memory=None):
    """DataLoader with the reference's settings (5 workers, pinned memory; :313-316)."""
    if pin_memory is None:
        pin_memory = torch.cuda.is_available()
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler,
                                       num_workers=num_workers, pin_memory=pin_memory,
                                       persistent_workers=num_workers > 0)
