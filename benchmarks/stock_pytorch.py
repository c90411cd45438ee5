#!/usr/bin/env python
"""Stock PyTorch-ROCm yard-stick: torch.nn ResNet-50, bf16 autocast, channels_last, MIOpen
convolutions, ``torch.optim.SGD`` — the "number to beat" of BASELINE.md (the reference
publishes none).  Same timing protocol as ``bench.py``.  Single GPU.

    python benchmarks/stock_pytorch.py --steps 30 --warmup 10 --batch 256 [--pure-bf16]
"""
import argparse
import json
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    def __init__(self, cin, f, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, f, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(f)
        self.conv2 = nn.Conv2d(f, f, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(f)
        self.conv3 = nn.Conv2d(f, 4 * f, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * f)
        self.down = None
        if project:
            self.down = nn.Sequential(nn.Conv2d(cin, 4 * f, 1, stride, bias=False),
                                      nn.BatchNorm2d(4 * f))

    def forward(self, x):
        s = x if self.down is None else self.down(x)
        o = F.relu(self.bn1(self.conv1(x)), inplace=True)
        o = F.relu(self.bn2(self.conv2(o)), inplace=True)
        return F.relu(self.bn3(self.conv3(o)) + s, inplace=True)


class ResNet50(nn.Module):
    def __init__(self, classes=1000):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64),
                                  nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
        blocks, cin = [], 64
        for gi, (f, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
            for bi in range(n):
                blocks.append(Bottleneck(cin, f, (2 if gi > 0 and bi == 0 else 1), bi == 0))
                cin = 4 * f
        self.blocks = nn.Sequential(*blocks)
        self.fc = nn.Linear(2048, classes)

    def forward(self, x):
        x = self.blocks(self.stem(x))
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--pure-bf16", action="store_true",
                    help="bf16 parameters instead of fp32 parameters under autocast")
    ap.add_argument("--autotune", action="store_true",
                    help="cudnn.benchmark=True as in the reference (:319); MIOpen's exhaustive "
                         "find can take minutes on a fresh box")
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.backends.cudnn.benchmark = bool(args.autotune)
    import threading

    stop = threading.Event()

    def heartbeat():  # MIOpen find mode is silent for minutes; keep the log moving
        n = 0
        while not stop.wait(30):
            n += 30
            print("[stock_pytorch] still running (%d s)" % n, flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()
    torch.manual_seed(42)
    model = ResNet50().to(dev).to(memory_format=torch.channels_last)
    if args.pure_bf16:
        model = model.to(torch.bfloat16)
    opt = torch.optim.SGD(model.parameters(), lr=0.001, momentum=0.9)
    x = torch.rand(args.batch, 3, 224, 224, device=dev).to(memory_format=torch.channels_last)
    if args.pure_bf16:
        x = x.to(torch.bfloat16)
    y = torch.randint(0, 1000, (args.batch,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        if args.pure_bf16:
            loss = F.cross_entropy(model(x).float(), y)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(x)
            loss = F.cross_entropy(out.float(), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    stop.set()
    print(json.dumps({"metric": "stock PyTorch-ROCm ResNet-50 images/sec", "autotune": bool(args.autotune), "mode":
                      "pure_bf16" if args.pure_bf16 else "autocast_bf16",
                      "value": round(args.batch * args.steps / el, 2), "batch": args.batch,
                      "ms_per_step": round(el / args.steps * 1e3, 3),
                      "loss": round(float(loss), 4)}))


if __name__ == "__main__":
    main()
