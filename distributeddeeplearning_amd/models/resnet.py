"""ResNet v1.5 family (18/34/50/101/152/200) with torchvision-compatible ``state_dict``.

Architecture spec, taken from the only in-tree model of the reference,
``HorovodTF/src/resnet_model.py`` — identical topology to the torchvision ResNet-50 that the
PyTorch flavour instantiates (``imagenet_pytorch_horovod.py:323``):

* depth table ``:306-313``; stem 7×7/2 conv 64, explicit pad 3, no conv bias ``:102-109`` →
  BN+ReLU ``:262`` → 3×3/2 max-pool ``:264-266``;
* four block groups, filters 64/128/256/512, first-block strides 1/2/2/2 ``:269-284``;
  projection shortcut (1×1 conv stride s + BN, no ReLU) only in the first block of a group
  ``:177-186,226-232``;
* bottleneck 1×1(f) → BN-ReLU → 3×3(f, stride s) → BN-ReLU → 1×1(4f) → BN → relu(x + shortcut)
  with the stride on the 3×3 (v1.5) ``:188-204``; basic block ``:133-153``;
* head: global average pool ``:287-289`` → dense ``num_classes`` ``:293-296``;
* BN momentum 0.1 (decay 0.9), eps 1e-5 ``:10-11``; optional zero-init of each block's last
  BN gamma ``:198-200``; He-normal conv init ``:108``.

MI355X layout: activations are NHWC (``channels_last``) bf16, conv weights are stored KRSC
(``channels_last`` on the OIHW logical shape, so ``state_dict`` still presents torchvision's
``[Cout, Cin, R, S]``), BatchNorm parameters and statistics stay fp32.  Compute goes through
``ops.functional`` — hand-written HIP kernels on the GPU, plain PyTorch on the CPU.
"""
from __future__ import annotations


import torch
import torch.nn as nn

from ..ops import functional as ops

__all__ = ["ResNet", "resnet", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152",
           "resnet200", "DEPTHS", "Conv2d", "BatchNormAct", "Linear", "Bottleneck",
           "BasicBlock", "count_parameters"]

DEPTHS = {
    18: ("basic", [2, 2, 2, 2]),
    34: ("basic", [3, 4, 6, 3]),
    50: ("bottleneck", [3, 4, 6, 3]),
    101: ("bottleneck", [3, 4, 23, 3]),
    152: ("bottleneck", [3, 8, 36, 3]),
    200: ("bottleneck", [3, 24, 36, 3]),
}


class Conv2d(nn.Module):
    """Bias-free convolution with explicit symmetric padding; weight KRSC in memory."""

    def __init__(self, cin, cout, kernel, stride=1, padding=0):
        super().__init__()
        self.cin, self.cout, self.kernel, self.stride, self.padding = cin, cout, kernel, stride, padding
        w = torch.empty(cout, cin, kernel, kernel)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.stride, self.padding)

    def extra_repr(self):
        return "{cin}, {cout}, kernel={kernel}, stride={stride}, padding={padding}".format(**self.__dict__)


class BatchNormAct(nn.Module):
    """Training-mode BatchNorm fused with an optional residual add and ReLU.

    Parameter / buffer names match ``torch.nn.BatchNorm2d`` (``weight``, ``bias``,
    ``running_mean``, ``running_var``, ``num_batches_tracked``)."""

    def __init__(self, channels, relu=True, momentum=0.1, eps=1e-5, zero_init=False):
        super().__init__()
        self.channels, self.relu, self.momentum, self.eps = channels, relu, momentum, eps
        self.weight = nn.Parameter(torch.zeros(channels) if zero_init else torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, residual=None):
        if self.training:
            self.num_batches_tracked += 1
        return ops.batchnorm_act(x, self.weight, self.bias, self.running_mean,
                                 self.running_var, self.training, self.momentum, self.eps,
                                 self.relu, residual)

    def extra_repr(self):
        return "{channels}, relu={relu}, momentum={momentum}, eps={eps}".format(**self.__dict__)


# ResNet.forward bumps every BatchNorm's ``num_batches_tracked`` with ONE multi-tensor add instead
# of 53 tiny launches; blocks used on their own keep the per-layer increment.
_NBT_BATCHED = [False]


def _cba(conv, bn, x, residual=None):
    """conv -> BatchNorm(+residual)(+ReLU) through the fused functional op."""
    if bn.training and not _NBT_BATCHED[0]:
        bn.num_batches_tracked += 1
    return ops.conv_bn_act(x, conv.weight, conv.stride, conv.padding, bn.weight, bn.bias,
                           bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps,
                           bn.relu, residual)


def _unit(conv, bn):
    if bn.training and not _NBT_BATCHED[0]:
        bn.num_batches_tracked += 1
    return (conv.weight, conv.stride, conv.padding, bn.weight, bn.bias, bn.running_mean,
            bn.running_var)


def _block_spec(pairs, downsample):
    units = [_unit(c, b) for c, b in pairs]
    ds = None
    if downsample is not None:
        u = _unit(downsample[0], downsample[1])
        ds = (u[0], u[1]) + u[3:]
    return units, ds


def _block(x, pairs, downsample):
    """Run a residual block through ``ops.residual_block`` (one fused autograd node on GPU)."""
    bn0 = pairs[0][1]
    units, ds = _block_spec(pairs, downsample)
    return ops.residual_block(x, units, ds, bn0.training, bn0.momentum, bn0.eps)


def _has_hooks(m):
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or
                m._backward_pre_hooks)


class Linear(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin).normal_(0.0, 0.01))
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class _Downsample(nn.Sequential):
    """Projection shortcut; indices 0 (conv) and 1 (bn) match torchvision's Sequential."""

    def __init__(self, cin, cout, stride):
        super().__init__(Conv2d(cin, cout, 1, stride, 0), BatchNormAct(cout, relu=False))

    def forward(self, x):
        return _cba(self[0], self[1], x)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, filters, stride, project, zero_init_residual):
        super().__init__()
        self.conv1 = Conv2d(cin, filters, 1, 1, 0)
        self.bn1 = BatchNormAct(filters)
        self.conv2 = Conv2d(filters, filters, 3, stride, 1)
        self.bn2 = BatchNormAct(filters)
        self.conv3 = Conv2d(filters, filters * 4, 1, 1, 0)
        self.bn3 = BatchNormAct(filters * 4, relu=True, zero_init=zero_init_residual)
        self.downsample = _Downsample(cin, filters * 4, stride) if project else None

    def pairs(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]

    def forward(self, x):
        return _block(x, self.pairs(), self.downsample)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, filters, stride, project, zero_init_residual):
        super().__init__()
        self.conv1 = Conv2d(cin, filters, 3, stride, 1)
        self.bn1 = BatchNormAct(filters)
        self.conv2 = Conv2d(filters, filters, 3, 1, 1)
        self.bn2 = BatchNormAct(filters, relu=True, zero_init=zero_init_residual)
        self.downsample = _Downsample(cin, filters, stride) if project else None

    def pairs(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]

    def forward(self, x):
        return _block(x, self.pairs(), self.downsample)


class ResNet(nn.Module):
    def __init__(self, depth=50, num_classes=1000, zero_init_residual=False, in_channels=3):
        super().__init__()
        if depth not in DEPTHS:
            raise ValueError("unsupported ResNet depth %r (choose from %s)" % (depth, sorted(DEPTHS)))
        kind, layers = DEPTHS[depth]
        block = Bottleneck if kind == "bottleneck" else BasicBlock
        self.depth, self.num_classes = depth, num_classes
        self.conv1 = Conv2d(in_channels, 64, 7, 2, 3)
        self.bn1 = BatchNormAct(64)
        cin = 64
        for gi, (filters, n_blocks) in enumerate(zip((64, 128, 256, 512), layers)):
            stride = 1 if gi == 0 else 2
            blocks = []
            for bi in range(n_blocks):
                s = stride if bi == 0 else 1
                project = bi == 0 and (s != 1 or cin != filters * block.expansion)
                blocks.append(block(cin, filters, s, project, zero_init_residual))
                cin = filters * block.expansion
            setattr(self, "layer%d" % (gi + 1), nn.Sequential(*blocks))
        self.fc = Linear(cin, num_classes)
        self.feature_dim = cin
        self.chain_blocks = True  # see _trunk

    def _bump_counters(self):
        """Every BatchNorm's ``num_batches_tracked`` += 1: one in-tree kernel over a table of the
        counters' addresses on the HIP path (rebuilt when a buffer moved), one multi-tensor add else."""
        cache = getattr(self, "_nbt_cache", None)
        if cache is None or cache[0] is not self.bn1.num_batches_tracked:
            # (module.to() / load_state_dict(assign=True) replace buffer objects: the first one is the witness)
            counters = [m.num_batches_tracked for m in self.modules() if isinstance(m, BatchNormAct)]
            table = None
            if counters[0].is_cuda:
                from ..ops import kernels

                if kernels.enabled():
                    table = torch.tensor([c.data_ptr() for c in counters], dtype=torch.int64).to(counters[0].device)
            cache = self._nbt_cache = (counters[0], counters, table)
        if cache[2] is not None:
            from ..ops import kernels

            kernels.bump_counters(cache[2], len(cache[1]))
            return
        torch._foreach_add_(cache[1], 1)

    def forward(self, x):
        batched = self.training and not _NBT_BATCHED[0]
        if batched:
            self._bump_counters()
            _NBT_BATCHED[0] = True
        try:
            if _has_hooks(self.conv1) or _has_hooks(self.bn1):
                x = ops.maxpool3x3s2(_cba(self.conv1, self.bn1, x))
            else:
                conv, bn = self.conv1, self.bn1
                if bn.training and not _NBT_BATCHED[0]:
                    bn.num_batches_tracked += 1
                x = ops.stem(x, conv.weight, conv.stride, conv.padding, bn.weight, bn.bias,
                             bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps)
            x = self._trunk(x)
            return self.fc(ops.global_avgpool(x))
        finally:
            if batched:
                _NBT_BATCHED[0] = False

    def _trunk(self, x):
        """The 4 block groups.  Without module hooks on them the 16 (ResNet-50) blocks run as one
        ``ops.residual_chain`` — no intermediate activation escapes, so the fused block nodes
        may hand each other ReLU-masked gradients; with hooks, module by module."""
        layers = (self.layer1, self.layer2, self.layer3, self.layer4)
        blocks = [b for layer in layers for b in layer]
        if not self.chain_blocks or any(_has_hooks(m) for m in blocks) or \
                any(_has_hooks(m) for m in layers):
            for layer in layers:
                x = layer(x)
            return x
        bn0 = blocks[0].bn1
        specs = [_block_spec(b.pairs(), b.downsample) for b in blocks]
        return ops.residual_chain(x, specs, bn0.training, bn0.momentum, bn0.eps)

    def to_compute(self, device, dtype=torch.bfloat16):
        """Move to ``device`` with conv/fc weights in ``dtype`` (KRSC) and BatchNorm + fc bias
        in fp32 — the mixed layout the HIP kernels expect."""
        self.to(device)
        for m in self.modules():
            if isinstance(m, Conv2d):
                m.weight.data = m.weight.data.to(dtype).contiguous(memory_format=torch.channels_last)
            elif isinstance(m, Linear):
                m.weight.data = m.weight.data.to(dtype)
        return self


def resnet(depth, **kw):
    return ResNet(depth, **kw)


def resnet18(**kw):
    return ResNet(18, **kw)


def resnet34(**kw):
    return ResNet(34, **kw)


def resnet50(**kw):
    return ResNet(50, **kw)


def resnet101(**kw):
    return ResNet(101, **kw)


def resnet152(**kw):
    return ResNet(152, **kw)


def resnet200(**kw):
    return ResNet(200, **kw)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters())
