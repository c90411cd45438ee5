"""cProfile of the host side of a training step (where does the issue time go).

The backward pass runs on autograd's own thread, which a profiler enabled on the main thread does
not see: every autograd node of ops/hip_ops.py gets its backward wrapped so that a second profiler
is switched on inside that thread for the duration of the call.
  python scripts/host_profile.py --model resnet152 --batch 16 --steps 12 --warmup 4
"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + (sys.argv[1:] or ["--steps", "20", "--warmup", "5"])
import torch
import bench
from distributeddeeplearning_amd.ops import hip_ops

bw = cProfile.Profile()


def _wrap(fn):
    def inner(*a, **k):
        bw.enable()
        try:
            return fn(*a, **k)
        finally:
            bw.disable()
    return staticmethod(inner)


for name in dir(hip_ops):
    obj = getattr(hip_ops, name)
    if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
        obj.backward = _wrap(obj.backward)

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
for title, p, n in (("main thread", pr, 30), ("backward nodes (autograd thread)", bw, 60)):
    s = io.StringIO()
    pstats.Stats(p, stream=s).sort_stats("tottime").print_stats(n)
    print("=====", title)
    print(s.getvalue()[:12000])
