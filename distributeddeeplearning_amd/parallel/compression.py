"""Gradient wire-format compression (Horovod's ``Compression.{none,fp16}`` plus bf16).

Not used by the reference (all fp32) but part of the Horovod surface named in
SURVEY §2.3.  On MI355X the natural wire format is bf16: same exponent range as fp32, half the
xGMI bytes, and it is already the gradient dtype of the bf16 compute path, in which case
compression is the identity.
"""
from __future__ import annotations

import torch

__all__ = ["Compression", "Compressor", "NoneCompressor", "FP16Compressor", "BF16Compressor"]


class Compressor:
    @staticmethod
    def compress(tensor):
        raise NotImplementedError

    @staticmethod
    def decompress(tensor, ctx):
        raise NotImplementedError


class NoneCompressor(Compressor):
    @staticmethod
    def compress(tensor):
        return tensor, None

    @staticmethod
    def decompress(tensor, ctx):
        return tensor


class _CastCompressor(Compressor):
    wire_dtype = torch.float16

    @classmethod
    def compress(cls, tensor):
        if tensor.is_floating_point() and tensor.dtype != cls.wire_dtype \
                and tensor.element_size() > 2:
            return tensor.to(cls.wire_dtype), tensor.dtype
        return tensor, None

    @classmethod
    def decompress(cls, tensor, ctx):
        return tensor if ctx is None else tensor.to(ctx)


class FP16Compressor(_CastCompressor):
    wire_dtype = torch.float16


class BF16Compressor(_CastCompressor):
    wire_dtype = torch.bfloat16


class Compression:
    none = NoneCompressor
    fp16 = FP16Compressor
    bf16 = BF16Compressor
