"""ctypes bindings of the HIP kernel library (``_ddl_kernels.so``).

Every launcher is ``extern "C" int ddl_<op>(…pointers…, …ints…, hipStream_t stream)`` and
returns 0 on success or a negative code when its host-side shape/alignment checks fail —
shapes are validated *before* a kernel is launched so a bad call raises in Python instead of
faulting the GPU.  Launches go to the current PyTorch HIP stream.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import native

__all__ = ["enabled", "require", "lib", "stream", "check", "KernelError"]

_state = {"checked": False, "ok": False}


class KernelError(RuntimeError):
    pass


def enabled() -> bool:
    """True when the kernel library is built and a GPU is present."""
    if not _state["checked"]:
        _state["ok"] = (os.environ.get("DDL_OPS", "auto").lower() != "aten"
                        and native.kernels_available() and torch.cuda.is_available())
        _state["checked"] = True
    return _state["ok"]


def require() -> bool:
    """Like enabled() but *raises* on a GPU machine whose kernel library is missing, so that
    a silent PyTorch fallback can never pass for the native path."""
    if enabled():
        return True
    if torch.cuda.is_available() and os.environ.get("DDL_OPS", "auto").lower() != "aten":
        native.kernels_lib()  # raises NativeLibraryMissing with build instructions
    return False


def lib():
    return native.kernels_lib()


def stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def check(code: int, what: str) -> None:
    if code != 0:
        raise KernelError("%s failed with code %d" % (what, code))


# ---- thin typed wrappers (filled in as kernels land) ---------------------------------------
def philox_uniform_(out: torch.Tensor, seed: int) -> None:
    """Fill a bf16/fp32 tensor with uniform [0,1) values (K19)."""
    assert out.is_contiguous() and out.is_cuda
    kind = {torch.bfloat16: 0, torch.float32: 1}[out.dtype]
    check(lib().ddl_philox_uniform(ptr(out), ctypes.c_int64(out.numel()), ctypes.c_int(kind),
                                   ctypes.c_uint64(seed), stream()), "philox_uniform")


def philox_labels_(out: torch.Tensor, n_classes: int, seed: int) -> None:
    assert out.is_contiguous() and out.is_cuda and out.dtype == torch.int64
    check(lib().ddl_philox_labels(ptr(out), ctypes.c_int64(out.numel()),
                                  ctypes.c_int(n_classes), ctypes.c_uint64(seed), stream()),
          "philox_labels")


def _i64(v):
    return ctypes.c_int64(int(v))


def _f32(v):
    return ctypes.c_float(float(v))


def fused_sgd_(w, g, v, lowp, lr, mu, wd, nesterov) -> None:
    """K14: one launch updates fp32 weights/momentum (+ bf16 shadow) over a flat range."""
    n = w.numel()
    assert w.dtype == torch.float32 and v.dtype == torch.float32 and v.numel() == n
    assert g.numel() == n and g.dtype in (torch.float32, torch.bfloat16)
    assert lowp is None or (lowp.dtype == torch.bfloat16 and lowp.numel() == n)
    check(lib().ddl_fused_sgd(ptr(w), ptr(g), ptr(v), ptr(lowp), _i64(n),
                              ctypes.c_int(int(g.dtype == torch.bfloat16)), _f32(lr), _f32(mu),
                              _f32(wd), ctypes.c_int(int(bool(nesterov))), stream()), "fused_sgd")


def segmented_norms(w, g, seg_offsets, seg_ends):
    """K15 pass 1: (‖w_k‖, ‖g_k‖) for every segment k of a flat run."""
    nseg = seg_ends.numel()
    out = torch.empty((2, nseg), dtype=torch.float32, device=w.device)
    check(lib().ddl_segmented_sqnorms(ptr(w), ptr(g), ctypes.c_int(int(g.dtype == torch.bfloat16)),
                                      ptr(seg_offsets), ptr(seg_ends), ctypes.c_int(nseg),
                                      _i64(w.numel()), ptr(out), stream()), "segmented_sqnorms")
    out = out.sqrt_()
    return out[0], out[1]


def fused_lars_(w, g, v, lowp, ratios, seg_offsets, seg_ends, lr, mu, wd) -> None:
    """K15 pass 2: v = mu*v + lr*ratio[k]*(g + wd*w); w -= v."""
    nseg = seg_ends.numel()
    assert ratios.dtype == torch.float32 and ratios.numel() == nseg and ratios.is_contiguous()
    check(lib().ddl_fused_lars(ptr(w), ptr(g), ptr(v), ptr(lowp), ptr(ratios), ptr(seg_offsets),
                               ptr(seg_ends), ctypes.c_int(nseg), _i64(w.numel()),
                               ctypes.c_int(int(g.dtype == torch.bfloat16)), _f32(lr), _f32(mu),
                               _f32(wd), stream()), "fused_lars")


def nchw_to_nhwc_normalize(x, mean=None, sd=None, in_scale=1.0):
    """K20: NCHW fp32/uint8 → NHWC bf16 with (x*in_scale - mean)/sd fused; returns a
    [B,C,H,W]-shaped channels_last bf16 tensor."""
    assert x.is_cuda and x.is_contiguous() and x.dim() == 4
    B, C, H, W = x.shape
    kind = {torch.float32: 0, torch.uint8: 1}[x.dtype]
    out = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=x.device)
    arr = (ctypes.c_float * 4)
    m = arr(*(list(mean) + [0.0] * (4 - C))) if mean is not None else None
    s = arr(*(list(sd) + [1.0] * (4 - C))) if sd is not None else None
    check(lib().ddl_nchw_to_nhwc_norm(ptr(x), ptr(out), _i64(B), ctypes.c_int(C), ctypes.c_int(H),
                                      ctypes.c_int(W), ctypes.c_int(kind), _f32(in_scale), m, s,
                                      stream()), "nchw_to_nhwc_norm")
    return out.permute(0, 3, 1, 2)
