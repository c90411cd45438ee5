"""Cycle stamps of the conv kernel's phases (library built with DDL_EXTRA_HIPCC_FLAGS=-DDDL_CONV_PROBES=1|2)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributeddeeplearning_amd.ops import hip_ops, kernels

kernels.require()
lib = kernels.lib()
lib.ddl_conv_set_debug_buffer.argtypes = [ctypes.c_void_p]
lib.ddl_conv_set_debug_buffer.restype = None
dev, bf, cl = "cuda", torch.bfloat16, torch.channels_last
SHAPES = {"c4": (3, 256, 256, 1, 14), "d1": (1, 1024, 512, 1, 14), "c3": (1, 1024, 256, 1, 14), "c5": (1, 256, 1024, 1, 14),
          "b4": (3, 128, 128, 1, 28), "a3": (3, 64, 64, 1, 56)}
B = 256
names = ["prologue", "first tile", "main loop", "epilogue", "mid: fetch", "mid: compute", "mid: wait+stash", "mid: barrier"]
for sid in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["c4", "d1", "c5"]):
    k, cin, cout, stride, hin = SHAPES[sid]
    pad = k // 2
    x = torch.randn((B, cin, hin, hin), device=dev).to(bf).contiguous(memory_format=cl)
    w = (torch.randn((cout, cin, k, k), device=dev) / math.sqrt(cin * k * k)).to(bf).contiguous(memory_format=cl)
    hout = (hin + 2 * pad - k) // stride + 1
    M = B * hout * hout
    y = torch.empty((B, cout, hout, hout), dtype=bf, device=dev, memory_format=cl)
    stats = torch.empty((kernels.conv_tiles_m(M), 2, cout), dtype=torch.float32, device=dev)
    dbg = torch.zeros((8192, 8), dtype=torch.int64, device=dev)
    geom = hip_ops._geom(x)
    for it in range(3):
        lib.ddl_conv_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr() if it == 2 else 0))
        kernels.conv_igemm(x, geom, w, y, cout, k, k, stride, pad, pad, 1, hout, hout, stats=stats)
    torch.cuda.synchronize()
    lib.ddl_conv_set_debug_buffer(ctypes.c_void_p(0))
    d = dbg.cpu()
    used = d[d[:, 2] > 0].double()
    print("%s: %d blocks; mean cycles per phase (100 MHz ticks x ? - see ratio):" % (sid, used.shape[0]))
    tot = used[:, :4].sum(1).mean()
    for i, n in enumerate(names):
        print("   %-16s %10.0f  (%.1f %% of block time)" % (n, used[:, i].mean(), 100 * used[:, i].mean() / tot))
