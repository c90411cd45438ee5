#!/usr/bin/env python
"""Cycles per phase of the row-streaming 3x3 kernel (needs a library built with -DDDL_CROWS_PROBE, loaded
through DDL_KERNELS_SO): multiply, DMA wait, barrier, DMA issue, epilogue - per step and wave."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import kernels as K  # noqa: E402

K.require()
DEV, BF, CL = torch.device("cuda", 0), torch.bfloat16, torch.channels_last
B, H = 256, 56
x = torch.randn((B, 64, H, 56), device=DEV).to(BF).contiguous(memory_format=CL)
w = (torch.randn((64, 64, 3, 3), device=DEV) / 24.0).to(BF).contiguous(memory_format=CL)
yp = torch.randn((B, 64, H, 56), device=DEV).to(BF).contiguous(memory_format=CL)
fco = torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3]).contiguous()
y = torch.empty_like(x)
names = ["multiply+epilogue", "dma wait", "barrier", "dma issue", "-"]
for bwd in (False, True):
    st = torch.zeros((B, 2, 64), device=DEV)
    for _ in range(3):
        if bwd:
            K.conv_rows3x3(x, w, y, stats=st, ep_y=yp, ep_fcoef=fco)
        else:
            K.conv_rows3x3(x, w, y, stats=st)
    torch.cuda.synchronize()
    t = st.view(B, 16, 8)[:, :8, :5].double() / (H // 2)      # cycles per step: [image][wave][phase]
    print("backward-data" if bwd else "forward", "- cycles per step (100 MHz counter ticks x ?), mean over images")
    for wv in range(8):
        print("  wave %d: " % wv + "  ".join("%s %7.1f" % (n, float(t[:, wv, i].mean())) for i, n in enumerate(names)))
    tt = st.view(B, 16, 8)[:, :8, 5:7].double()
    print("  prologue %.0f cycles, whole kernel %.0f cycles (per workgroup, mean)" % (float(tt[:, :, 0].mean()), float(tt[:, :, 1].mean())))
    print("  all  : " + "  ".join("%s %7.1f" % (n, float(t[:, :, i].mean())) for i, n in enumerate(names)) +
          "   sum %.1f" % float(t.sum(2).mean()))
