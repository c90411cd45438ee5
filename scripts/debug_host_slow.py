"""Why is a configuration host-bound?  Per-phase host times + allocator counters per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributeddeeplearning_amd.models import build_model
from distributeddeeplearning_amd.ops import functional as ops
from distributeddeeplearning_amd.optim import FusedSGD

name = sys.argv[1] if len(sys.argv) > 1 else "resnet34"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda")
model = build_model(name, num_classes=1000)
model.to_compute(dev, torch.bfloat16)
model.train()
opt = FusedSGD(model.named_parameters(), lr=0.001, momentum=0.9)
x = torch.rand((batch, 3, 224, 224), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 1000, (batch,), device=dev)
for step in range(8):
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    opt.zero_grad()
    out = model(x)
    loss = ops.softmax_cross_entropy(out, t)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    ms1 = torch.cuda.memory_stats()
    print("step %d host fwd %.1f bwd %.1f opt %.1f sync %.1f ms | segments +%d retries +%d reserved %.1f GB" % (
        step, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3,
        ms1["segment.all.allocated"] - ms0["segment.all.allocated"],
        ms1["num_alloc_retries"] - ms0["num_alloc_retries"], ms1["reserved_bytes.all.current"] / 2**30), flush=True)
