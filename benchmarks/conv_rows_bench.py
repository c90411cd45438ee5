#!/usr/bin/env python
"""a3 (64 -> 64 channels, 3x3, 56 x 56, batch 256) forward + statistics and masked backward-data:
row-streaming kernel (DDL_CONV_ROWS=1) or the tile kernel (DDL_CONV_ROWS=0), one process each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import hip_ops as hip, kernels as K  # noqa: E402

K.require()
DEV, BF, CL = torch.device("cuda", 0), torch.bfloat16, torch.channels_last


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 256
x = torch.randn((B, 64, 56, 56), device=DEV).to(BF).contiguous(memory_format=CL)
w = (torch.randn((64, 64, 3, 3), device=DEV) / 24.0).to(BF).contiguous(memory_format=CL)
yp = torch.randn((B, 64, 56, 56), device=DEV).to(BF).contiguous(memory_format=CL)
fco = torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3]).contiguous()
y_out, z_out, st = torch.empty_like(x), torch.empty_like(x), torch.empty((B, 2, 64), device=DEV)
if K.conv_rows3x3_ok(64, 64, 3, 3, 1, 1, 56, 56):
    xf = timeit(lambda: K.conv_rows3x3(x, w, y_out, stats=st, in_fcoef=fco, z_out=z_out))
    ap = timeit(lambda: K.bn_apply_fwd(x, fco[0], fco[1], None, z_out, B * 56 * 56, 64, True))
    print("forward with the BatchNorm apply of the layer in front inside: %.1f us; that apply pass alone: %.1f us" % (xf, ap), flush=True)
with torch.no_grad():
    f = timeit(lambda: hip._conv_fwd(x, w, 1, 1, True))
    b = timeit(lambda: hip._conv_dgrad_bn(x, w, x.shape, 1, 1, yp, fco.data_ptr()))
print("DDL_CONV_ROWS=%s: forward+stats %.1f us, masked backward-data %.1f us" % (os.environ.get("DDL_CONV_ROWS", "1"), f, b), flush=True)
