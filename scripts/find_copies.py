"""Which host call sites issue device-to-device memcpys during a flagship step (torch profiler)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

sys.argv = ["bench.py", "--steps", "3", "--warmup", "3"]
import bench

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.main()

cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::empty_strided"):
        frames = [f for f in (ev.stack or []) if "distributeddeeplearning_amd" in f or "bench.py" in f]
        cnt[(ev.name, frames[0] if frames else "?")] += 1
for (name, site), n in cnt.most_common(40):
    print("%5d  %-20s %s" % (n, name, site))
names = collections.Counter(ev.name for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
print("--- device-side events")
for k, v in names.most_common(12):
    print("%5d  %s" % (v, k[:110]))
