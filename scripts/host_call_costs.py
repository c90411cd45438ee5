"""Host cost per call (us) of the pieces a launch is made of, on tiny tensors so the GPU never
throttles the host: ctypes launcher wrappers, the bare ctypes call, torch.empty, event pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributeddeeplearning_amd.ops import kernels as K, hip_ops

K.require()
dev, bf = "cuda", torch.bfloat16
N = 4000


def cost(fn, n=N):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / n * 1e6


M, C = 256, 64
y = torch.randn(M, C, device=dev).to(bf)
z = torch.empty_like(y)
sc, sh = torch.rand(C, device=dev), torch.rand(C, device=dev)
print("bn_apply_fwd wrapper            %6.2f us" % cost(lambda: K.bn_apply_fwd(y, sc, sh, None, z, M, C, True)))
x = torch.randn(2, 64, 8, 8, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 64, 1, 1, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
out = torch.empty_like(x)
g = hip_ops._geom(x)
print("conv_igemm wrapper              %6.2f us" % cost(lambda: K.conv_igemm(x, g, w, out, 64, 1, 1, 1, 0, 0, 1, 8, 8)))
print("_geom(x)                        %6.2f us" % cost(lambda: hip_ops._geom(x)))
print("torch.empty (64 KB)             %6.2f us" % cost(lambda: torch.empty(32768, dtype=bf, device=dev)))
print("torch.empty_like                %6.2f us" % cost(lambda: torch.empty_like(x)))
print("x.data_ptr() x 5                %6.2f us" % cost(lambda: (x.data_ptr(), w.data_ptr(), out.data_ptr(), y.data_ptr(), z.data_ptr())))
print("K._raw_stream()                 %6.2f us" % cost(K._raw_stream))
side = hip_ops._side_stream(torch.device("cuda", 0))
raw, ev = side.cuda_stream, hip_ops._SIDE["event"]
print("stream_wait_stream (event pair) %6.2f us" % cost(lambda: K.stream_wait_stream(raw, K._raw_stream(), ev)))
f = K._fn("ddl_bn_apply_fwd_ex2", None) if False else None
import ctypes
lib = K.lib()
fn = lib.ddl_stream_priority_range
a, b = ctypes.c_int(0), ctypes.c_int(0)
ra, rb = ctypes.byref(a), ctypes.byref(b)
print("bare ctypes call (2 args, no launch) %6.2f us" % cost(lambda: fn(ra, rb)))
t = torch.zeros(8, device=dev)
print("torch add_ (ATen launch)        %6.2f us" % cost(lambda: t.add_(1.0)))
