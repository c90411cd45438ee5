"""Functional op surface used by the models.

Each op has two implementations:

* the **HIP path** (``ops/hip_ops.py`` → ``_ddl_kernels.so``): hand-written gfx950 kernels on
  NHWC bf16 tensors, wrapped in ``torch.autograd.Function``s.  This is the path that runs on
  MI355X; it fails loudly if the kernel library is missing.
* the **reference path** (this file, plain PyTorch): used on CPU (tests, the gloo plumbing
  configuration of BASELINE.json) and as the fp32 numerical oracle for every kernel test.
  ``DDL_OPS=aten`` forces it on the GPU too, which is how the stock-PyTorch/MIOpen baseline
  in BASELINE.md is measured.

Semantics follow the reference's model code: conv without bias and explicit padding
(``HorovodTF/src/resnet_model.py:84-109``), training-mode BatchNorm with momentum 0.1
(= decay 0.9) and eps 1e-5 (``:10-11,40-49``) followed by ReLU, with the residual form
``relu(bn(x) + shortcut)`` (``:201-204``), 3×3/2 max-pool (``:264-266``), 7×7 global average
pool (``:287-289``), dense head (``:293-296``), mean softmax cross-entropy
(``imagenet_pytorch_horovod.py:340``).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

__all__ = ["use_hip", "conv2d", "conv_bn_act", "residual_block", "batchnorm_act", "maxpool3x3s2", "global_avgpool", "linear",
           "softmax_cross_entropy", "topk_correct", "set_backend", "backend"]

_FORCED = os.environ.get("DDL_OPS", "auto").lower()  # auto | hip | aten


def set_backend(name: str) -> None:
    global _FORCED
    if name not in ("auto", "hip", "aten"):
        raise ValueError(name)
    _FORCED = name


def backend() -> str:
    return _FORCED


def use_hip(t: torch.Tensor) -> bool:
    """HIP kernels serve CUDA(HIP)-device bf16 tensors unless DDL_OPS=aten."""
    if _FORCED == "aten" or not t.is_cuda:
        return False
    if t.dtype != torch.bfloat16:
        if _FORCED == "hip":
            raise RuntimeError("DDL_OPS=hip requires bf16 activations, got %s" % t.dtype)
        return False
    from . import kernels

    return kernels.require()


# ------------------------------------------------------------------------------------------
def conv2d(x, weight, stride=1, padding=0):
    if use_hip(x):
        from . import hip_ops

        if hip_ops.conv_supported(weight, stride, padding):
            return hip_ops.conv2d(x, weight, stride, padding)
        _warn_once("conv2d", "Cin=%d Cout=%d %dx%d" % (weight.shape[1], weight.shape[0],
                                                        weight.shape[2], weight.shape[3]))
    return F.conv2d(x, weight.to(x.dtype), None, stride, padding)


_warned = set()


def _warn_once(op, what):
    """Shapes outside the HIP kernels' envelope run on ATen — loudly, never silently."""
    key = (op, what)
    if key not in _warned:
        _warned.add(key)
        import warnings

        warnings.warn("ops.%s: no HIP kernel for %s; using the ATen implementation" % (op, what))


def conv_bn_act(x, weight, stride, padding, gamma, beta, running_mean, running_var, training,
                momentum=0.1, eps=1e-5, relu=True, residual=None):
    """``act(bn(conv(x)) [+ residual])`` — on the HIP path the convolution's epilogue emits the
    per-channel batch-statistic partial sums, so BatchNorm never re-reads the conv output to
    compute them (SURVEY §2.6 K1 note / K6)."""
    if use_hip(x):
        from . import hip_ops

        if hip_ops.conv_supported(weight, stride, padding) and hip_ops.bn_supported(weight.shape[0]):
            y, stats = hip_ops.conv2d_stats(x, weight, stride, padding, bool(training))
            return hip_ops.batchnorm_act(y, gamma, beta, running_mean, running_var, training,
                                         momentum, eps, relu, residual, stats)
    y = conv2d(x, weight, stride, padding)
    return batchnorm_act(y, gamma, beta, running_mean, running_var, training, momentum, eps, relu,
                         residual)


def residual_block(x, units, downsample, training, momentum=0.1, eps=1e-5):
    """A whole residual block: ``units`` is the main path as a list of
    ``(weight, stride, padding, gamma, beta, running_mean, running_var)`` conv-BN-ReLU pairs (the
    last one adds the shortcut before its ReLU), ``downsample`` is None (identity shortcut) or
    ``(weight, stride, gamma, beta, running_mean, running_var)`` for the 1x1 projection
    (``HorovodTF/src/resnet_model.py:112-204``).  On the HIP path the block is one autograd
    node with a hand-scheduled backward (``hip_ops.ResidualBlockFn``)."""
    if use_hip(x):
        from . import hip_ops

        convs = [u[0] for u in units] + ([downsample[0]] if downsample is not None else [])
        if hip_ops.residual_block_supported(convs):
            return hip_ops.residual_block(x, units, downsample, training, momentum, eps)
    shortcut = x
    if downsample is not None:
        w, s, g, b, rm, rv = downsample
        shortcut = conv_bn_act(x, w, s, 0, g, b, rm, rv, training, momentum, eps, False)
    cur = x
    for i, (w, s, p, g, b, rm, rv) in enumerate(units):
        last = i == len(units) - 1
        cur = conv_bn_act(cur, w, s, p, g, b, rm, rv, training, momentum, eps, True,
                          shortcut if last else None)
    return cur


def stem(x, weight, stride, padding, gamma, beta, running_mean, running_var, training, momentum=0.1,
         eps=1e-5):
    """ResNet stem: conv -> BatchNorm -> ReLU -> 3x3/2 max pool (``HorovodTF/src/resnet_model.py:
    256-266``).  In training mode on the HIP path this is one fused node (``hip_ops.StemFn``);
    otherwise the composition of the three ops."""
    if training and use_hip(x):
        from . import hip_ops

        if hip_ops.stem_supported(weight, stride, padding) and not x.requires_grad:
            return hip_ops.stem(x, weight, gamma, beta, running_mean, running_var, momentum, eps)
    y = conv_bn_act(x, weight, stride, padding, gamma, beta, running_mean, running_var, training,
                    momentum, eps, True)
    return maxpool3x3s2(y)


def residual_chain(x, blocks, training, momentum=0.1, eps=1e-5):
    """Consecutive residual blocks, ``blocks`` = list of ``(units, downsample)`` as taken by
    :func:`residual_block`.  The intermediate activations are not exposed, which lets the HIP
    path pass ReLU-masked gradients between the fused block nodes
    (``hip_ops.residual_chain``); elsewhere it is just the loop over the blocks."""
    if use_hip(x):
        from . import hip_ops

        if all(hip_ops.residual_block_supported(
                [u[0] for u in units] + ([ds[0]] if ds is not None else []))
               for units, ds in blocks):
            return hip_ops.residual_chain(x, blocks, training, momentum, eps)
    for units, ds in blocks:
        x = residual_block(x, units, ds, training, momentum, eps)
    return x


def batchnorm_act(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5,
                  relu=True, residual=None):
    """``act(bn(x) [+ residual])`` with act = ReLU or identity; running stats updated in place
    when training (unbiased variance, like torch)."""
    if use_hip(x):
        from . import hip_ops

        if hip_ops.bn_supported(x.shape[1]):
            return hip_ops.batchnorm_act(x, gamma, beta, running_mean, running_var, training,
                                         momentum, eps, relu, residual)
        _warn_once("batchnorm_act", "C=%d" % x.shape[1])
    y = F.batch_norm(x.float() if x.dtype != torch.float32 and not x.is_cuda else x,
                     running_mean, running_var, gamma, beta, training, momentum, eps)
    y = y.to(x.dtype)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def maxpool3x3s2(x):
    if use_hip(x):
        from . import hip_ops

        return hip_ops.maxpool3x3s2(x)
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def global_avgpool(x):
    """[B,C,H,W] → [B,C]."""
    if use_hip(x):
        from . import hip_ops

        return hip_ops.global_avgpool(x)
    return x.mean(dim=(2, 3))


def linear(x, weight, bias=None):
    if use_hip(x):
        from . import hip_ops

        if x.dim() == 2 and hip_ops.linear_supported(weight):
            return hip_ops.linear(x, weight, bias)
        _warn_once("linear", "in=%d out=%d" % (weight.shape[1], weight.shape[0]))
    return F.linear(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))


def softmax_cross_entropy(logits, target):
    """Mean softmax cross-entropy, computed in fp32; returns an fp32 scalar."""
    if logits.is_cuda and _FORCED != "aten":
        from . import kernels

        if kernels.require():
            from . import hip_ops

            return hip_ops.softmax_cross_entropy(logits, target)
    return F.cross_entropy(logits.float(), target)


def topk_correct(logits, target, ks=(1, 5)):
    """Number of samples whose label is within the top-k logits, for each k (eval metric;
    ``imagenet_keras_horovod.py:307`` accuracy + top_k_categorical_accuracy)."""
    if use_hip(logits) and logits.dim() == 2 and logits.dtype == torch.bfloat16 and \
            logits.stride(1) == 1:
        from . import kernels as K

        rank = K.label_rank(logits, target.contiguous())  # one wave per row, no sort
        return [int((rank < k).sum().item()) for k in ks]
    maxk = max(ks)
    _, pred = logits.float().topk(maxk, dim=1)
    hit = pred.eq(target.view(-1, 1))
    return [int(hit[:, :k].any(dim=1).sum().item()) for k in ks]
