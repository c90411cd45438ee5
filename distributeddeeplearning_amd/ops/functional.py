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
           "softmax_cross_entropy", "topk_correct", "topk_hits", "NoHipKernel", "set_backend", "backend"]

_FORCED = os.environ.get("DDL_OPS", "auto").lower()  # auto | hip | aten


def set_backend(name: str) -> None:
    global _FORCED
    if name not in ("auto", "hip", "aten"):
        raise ValueError(name)
    _FORCED = name


def backend() -> str:
    return _FORCED


class NoHipKernel(RuntimeError):
    """A GPU tensor reached an op the HIP kernels do not serve.  There is ONE code path on the GPU:
    nothing falls back to ATen/MIOpen unless ``DDL_OPS=aten`` asks for the oracle explicitly."""


def use_hip(t: torch.Tensor) -> bool:
    """HIP kernels serve every GPU tensor (bf16 activations) unless ``DDL_OPS=aten`` selects the
    PyTorch oracle; CPU tensors always take the PyTorch path."""
    if _FORCED == "aten" or not t.is_cuda:
        return False
    if t.dtype != torch.bfloat16:
        raise NoHipKernel("the HIP kernels compute in bf16, got a %s tensor on %s: move the model with "
                          ".to_compute(device, torch.bfloat16) and feed bf16 batches, or set DDL_OPS=aten "
                          "for the PyTorch oracle" % (t.dtype, t.device))
    from . import kernels

    return kernels.require()


def _no_kernel(op, what):
    raise NoHipKernel("ops.%s: no HIP kernel for %s (set DDL_OPS=aten to run the PyTorch oracle instead)"
                      % (op, what))


# ------------------------------------------------------------------------------------------
def conv2d(x, weight, stride=1, padding=0):
    if use_hip(x):
        from . import hip_ops

        if hip_ops.conv_supported(weight, stride, padding):
            return hip_ops.conv2d(x, weight, stride, padding)
        _no_kernel("conv2d", "Cin=%d Cout=%d %dx%d stride %s" % (weight.shape[1], weight.shape[0],
                                                                  weight.shape[2], weight.shape[3], stride))
    return F.conv2d(x, weight.to(x.dtype), None, stride, padding)


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
        _no_kernel("conv_bn_act", "weight %s stride %s" % (tuple(weight.shape), stride))
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
        _no_kernel("residual_block", "convolutions %s" % [tuple(w.shape) for w in convs])
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
        _no_kernel("residual_chain", "one of the blocks' convolutions")
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
        _no_kernel("batchnorm_act", "C=%d" % x.shape[1])
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
        _no_kernel("linear", "in=%d out=%d on a %d-d input" % (weight.shape[1], weight.shape[0], x.dim()))
    return F.linear(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))


def softmax_cross_entropy(logits, target):
    """Mean softmax cross-entropy, computed in fp32; returns an fp32 scalar."""
    if logits.is_cuda and _FORCED != "aten":
        from . import kernels

        if kernels.require():
            from . import hip_ops

            return hip_ops.softmax_cross_entropy(logits, target)
    return F.cross_entropy(logits.float(), target)


def topk_hits(logits, target, ks=(1, 5)):
    """Per-k count of samples whose label is within the top-k logits, as a float64 tensor on the
    logits' device - no host synchronisation, so evaluation loops accumulate it and read once
    (eval metric: ``imagenet_keras_horovod.py:307`` accuracy + top_k_categorical_accuracy,
    ``imagenet_estimator_tf_horovod.py:209-213``).  On the GPU this is K21: one wave per row finds
    the rank of the label's logit, no sort and no [B, k] index tensor."""
    if logits.is_cuda and _FORCED != "aten":
        from . import kernels as K

        K.require()
        if logits.dim() != 2:
            _no_kernel("topk_hits", "a %d-d logits tensor" % logits.dim())
        lg = logits.detach()
        if lg.dtype != torch.bfloat16 or lg.stride(1) != 1:
            lg = lg.to(torch.bfloat16).contiguous()   # the classifier head emits bf16; fp32 callers are rounded once
        rank = K.label_rank(lg, target.contiguous())
        return torch.stack([(rank < k).sum() for k in ks]).double()
    maxk = max(ks)
    _, pred = logits.detach().float().topk(maxk, dim=1)
    hit = pred.eq(target.view(-1, 1))
    return torch.stack([hit[:, :k].any(dim=1).sum() for k in ks]).double()


def topk_correct(logits, target, ks=(1, 5)):
    """:func:`topk_hits` as Python ints (one device synchronisation)."""
    return [int(v) for v in topk_hits(logits, target, ks).tolist()]
