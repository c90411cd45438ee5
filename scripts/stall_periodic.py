"""Is the ~1.6 s stall tied to wall-clock time rather than to this framework?  A stock
torch.nn loop (bf16 matmuls + elementwise ops, no in-tree kernels) with the same watchdog."""
import sys, time, faulthandler
import torch
dev = torch.device("cuda")
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
t_start = time.perf_counter()
dur = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n = 0
stalls = []
while time.perf_counter() - t_start < dur:
    t0 = time.perf_counter()
    for _ in range(10):
        c = a @ b
        c = torch.relu(c)
        d = torch.empty_like(c)
        d.copy_(c)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n += 1
    if dt > 0.25:
        stalls.append((round(t0 - t_start, 2), round(dt, 3)))
print("stock loop: %d iterations in %.0f s, stalls (start s, length s): %s" % (n, dur, stalls))
