"""autograd wrappers around the HIP kernels (GPU path of ``ops.functional``).

Ops whose hand-written kernel has not landed yet call the ATen op explicitly here — never as
a hidden fallback: ``IMPLEMENTED`` lists which ops run on our own kernels, and
``tests/test_gpu_kernels.py`` asserts the flagship path uses them.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

IMPLEMENTED = set()


def conv2d(x, weight, stride, padding):
    return F.conv2d(x, weight, None, stride, padding)


def batchnorm_act(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu,
                  residual):
    y = F.batch_norm(x, running_mean, running_var, gamma, beta, training, momentum, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def maxpool3x3s2(x):
    return F.max_pool2d(x, 3, 2, 1)


def global_avgpool(x):
    return x.mean(dim=(2, 3))


def linear(x, weight, bias):
    return F.linear(x, weight, None if bias is None else bias.to(x.dtype))


def softmax_cross_entropy(logits, target):
    return F.cross_entropy(logits.float(), target)
