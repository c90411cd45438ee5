"""ctypes bindings of the HIP kernel library (``_ddl_kernels.so``).

Every launcher is ``extern "C" int ddl_<op>(…pointers…, …ints…, hipStream_t stream)`` and
returns 0 on success or a negative code when its host-side shape/alignment checks fail —
shapes are validated *before* a kernel is launched so a bad call raises in Python instead of
faulting the GPU.  Launches go to the current PyTorch HIP stream.
"""
from __future__ import annotations

import ctypes
import functools
import threading
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
    """The stream this thread's launchers queue on: torch's current stream unless ``launch_on`` redirects."""
    return ctypes.c_void_p(_raw_stream())


# ---- fast path for the launchers called ~500 times per training step ---------------------------
# With ``argtypes`` declared, ctypes converts plain Python ints / floats / None in C instead of
# through ~30 freshly built c_int / c_void_p objects per call (6.6 us -> 2.7 us per launch on the
# host, measured): the host has to stay ahead of ~50 us kernels in the late layers' backward.
_SIG = {"p": ctypes.c_void_p, "i": ctypes.c_int, "q": ctypes.c_longlong, "f": ctypes.c_float,
        "u": ctypes.c_uint, "Q": ctypes.c_ulonglong}
_bound = {}


def _fn(name, sig):
    f = _bound.get(name)
    if f is None:
        f = getattr(lib(), name)
        f.argtypes = [_SIG[c] for c in sig if c != " "]
        f.restype = ctypes.c_int
        _bound[name] = f
    return f


class _StreamOverride(threading.local):
    raw = None   # raw HIP stream that replaces "the current stream" for this thread's launchers (see launch_on)


_STREAM_OVERRIDE = _StreamOverride()


def _raw_stream():
    """Raw handle of the current stream of the current device (an int)."""
    o = _STREAM_OVERRIDE.raw
    if o is not None:
        return o
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class launch_on:
    """Every launcher this thread calls inside the block queues on raw stream ``raw`` instead of torch's
    current stream, while tensors keep being allocated from the current stream's pool (the caller orders
    the two streams with events; no allocator bookkeeping per tensor)."""

    def __init__(self, raw):
        self.raw = raw

    def __enter__(self):
        self.prev = _STREAM_OVERRIDE.raw
        _STREAM_OVERRIDE.raw = self.raw
        return self

    def __exit__(self, *exc):
        _STREAM_OVERRIDE.raw = self.prev
        return False


def _p(t):
    """Device address of a tensor; raw addresses (ints: slices of a per-layer scratch block, which
    cost a tensor object each when expressed as views) pass through."""
    if t is None:
        return None
    return t if t.__class__ is int else t.data_ptr()


def stream_priority_range():
    """(least, greatest) stream priority of the current device; numerically lower = higher."""
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().ddl_stream_priority_range(ctypes.byref(least), ctypes.byref(greatest)),
          "hipDeviceGetStreamPriorityRange")
    return least.value, greatest.value


def create_stream(priority: int) -> int:
    """A non-blocking HIP stream of the given priority on the current device (never destroyed:
    process lifetime); returns the raw handle for ``torch.cuda.ExternalStream``.  Created inside
    the kernel library so that it belongs to the HIP runtime instance the launches go through."""
    h = ctypes.c_void_p()
    check(lib().ddl_stream_create(ctypes.c_int(priority), ctypes.byref(h)), "hipStreamCreateWithPriority")
    return h.value


def create_event() -> int:
    """A timing-less HIP event (process lifetime) for :func:`stream_wait_stream`."""
    h = ctypes.c_void_p()
    check(lib().ddl_event_create(ctypes.byref(h)), "hipEventCreate")
    return h.value


def stream_wait_stream(waiter: int, producer: int, event: int) -> None:
    """Order raw stream ``waiter`` after the work queued on raw stream ``producer`` so far."""
    f = _fn("ddl_stream_wait_stream", "ppp")
    check(f(waiter, producer, event), "stream_wait_stream")


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


def fused_sgd_(w, g, v, lowp, lr, mu, wd, nesterov, lr_scale=None) -> None:
    """K14: one launch updates fp32 weights/momentum (+ bf16 shadow) over a flat range.
    ``lr_scale``: optional 1-element fp32 device tensor multiplied into ``lr`` on the device
    (learning-rate schedules under a captured HIP graph)."""
    n = w.numel()
    assert w.dtype == torch.float32 and v.dtype == torch.float32 and v.numel() == n
    assert g.numel() == n and g.dtype in (torch.float32, torch.bfloat16)
    assert lowp is None or (lowp.dtype == torch.bfloat16 and lowp.numel() == n)
    assert lr_scale is None or (lr_scale.dtype == torch.float32 and lr_scale.numel() == 1)
    check(lib().ddl_fused_sgd_ex(ptr(w), ptr(g), ptr(v), ptr(lowp), _i64(n),
                                 ctypes.c_int(int(g.dtype == torch.bfloat16)), _f32(lr), _f32(mu),
                                 _f32(wd), ctypes.c_int(int(bool(nesterov))), ptr(lr_scale),
                                 stream()), "fused_sgd")


@functools.lru_cache(maxsize=None)
def _sqnorm_scratch(n):
    lib().ddl_segmented_sqnorms_scratch.restype = ctypes.c_longlong
    return int(lib().ddl_segmented_sqnorms_scratch(_i64(n)))


def segmented_norms(w, g, seg_offsets, seg_ends):
    """K15 pass 1: (‖w_k‖, ‖g_k‖) for every segment k of a flat run."""
    nseg = seg_ends.numel()
    out = torch.empty((2, nseg), dtype=torch.float32, device=w.device)
    scratch = torch.empty(_sqnorm_scratch(w.numel()), dtype=torch.float32, device=w.device)
    check(lib().ddl_segmented_sqnorms(ptr(w), ptr(g), ctypes.c_int(int(g.dtype == torch.bfloat16)),
                                      ptr(seg_offsets), ptr(seg_ends), ctypes.c_int(nseg),
                                      _i64(w.numel()), ptr(out), ptr(scratch), stream()), "segmented_sqnorms")
    out = out.sqrt_()
    return out[0], out[1]


def lars_trust_ratios(w, g, seg_offsets, seg_ends, trust_coefficient, wd, eps):
    """K15 passes 1 + 1b without a framework operator: squared norms per segment, then
    ``tc*|w| / (|g| + wd*|w| + eps)`` (1 where a norm is zero) - an fp32 [nseg] tensor."""
    nseg = seg_ends.numel()
    buf = torch.empty(3 * nseg + _sqnorm_scratch(w.numel()), dtype=torch.float32, device=w.device)
    check(lib().ddl_segmented_sqnorms(ptr(w), ptr(g), ctypes.c_int(int(g.dtype == torch.bfloat16)),
                                      ptr(seg_offsets), ptr(seg_ends), ctypes.c_int(nseg),
                                      _i64(w.numel()), ptr(buf), ctypes.c_void_p(buf.data_ptr() + 12 * nseg),
                                      stream()), "segmented_sqnorms")
    check(lib().ddl_lars_trust_ratios(ptr(buf), ctypes.c_int(nseg), _f32(trust_coefficient), _f32(wd), _f32(eps),
                                      ctypes.c_void_p(buf.data_ptr() + 8 * nseg), stream()), "lars_trust_ratios")
    return buf[2 * nseg: 3 * nseg]


def fused_lars_(w, g, v, lowp, ratios, seg_offsets, seg_ends, lr, mu, wd, lr_scale=None) -> None:
    """K15 pass 2: v = mu*v + lr*ratio[k]*(g + wd*w); w -= v."""
    nseg = seg_ends.numel()
    assert ratios.dtype == torch.float32 and ratios.numel() == nseg and ratios.is_contiguous()
    assert lr_scale is None or (lr_scale.dtype == torch.float32 and lr_scale.numel() == 1)
    check(lib().ddl_fused_lars_ex(ptr(w), ptr(g), ptr(v), ptr(lowp), ptr(ratios), ptr(seg_offsets),
                                  ptr(seg_ends), ctypes.c_int(nseg), _i64(w.numel()),
                                  ctypes.c_int(int(g.dtype == torch.bfloat16)), _f32(lr), _f32(mu),
                                  _f32(wd), ptr(lr_scale), stream()), "fused_lars")


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


# ---- convolution / GEMM (csrc/kernels/conv_igemm.hip, conv_wgrad.hip) ------------------------
def _i32(v):
    return ctypes.c_int(int(v))


# timing probe (run through scripts/gpu_sweep.sh): DDL_PROBE_SKIP_WGRAD=all|late|early drops the weight-gradient GEMMs, =reduce the plain slab reductions
_PROBE_SKIP_WGRAD = os.environ.get("DDL_PROBE_SKIP_WGRAD", "")

# diagnostic: a list that receives one tuple per conv launch (scripts/conv_roofline.py); None = off
LAUNCH_LOG = None


def nhwc_dims(x):
    """(B, C, H, W, stride_b, stride_h, stride_w) of a logically-NCHW tensor stored NHWC."""
    assert x.dim() == 4
    B, C, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if C != 1 and sc != 1:
        raise KernelError("tensor is not channels_last (NHWC) in memory: shape %s stride %s"
                          % (tuple(x.shape), tuple(x.stride())))
    if H == 1 and W == 1:
        sh = sw = C  # degenerate spatial dims carry arbitrary strides
    return B, C, H, W, sb, sh, sw


def storage_numel_from(x):
    """Elements from x's first element to the end of its allocation (buffer range for loads)."""
    st = x.untyped_storage()
    return (st.nbytes() - x.storage_offset() * x.element_size()) // x.element_size()


def conv_igemm(x_ptr_tensor, geom, w, y, n_out, R, S, ostride, pad, pad_w, idil, Hout, Wout,
               addend=None, stats=None, addend_sub2=False, ep_y=None, ep_fcoef=None, ep_mask=None,
               ep_y2=None, stats2=None, ep_maskbits=None, wt_rows=0, x2=None, w2=None, c2=0, bias=None, stream=None, w_numel=None):
    """Raw launcher.  ``geom`` = (B, Hin, Win, C, xs_b, xs_h, xs_w, x_numel).

    ``wt_rows`` > 0: backward-data mode - ``w`` is the FORWARD filter ``[wt_rows][R][S][n_out]`` and
    the kernel transposes it on the way from LDS to the matrix cores (no transposed copy exists).
    ``x2`` [M, c2] / ``w2`` [n_out, c2] / ``bias`` [n_out] fp32 (plain-GEMM launches only): a second
    reduction segment and a per-channel constant, ``y = x w^T + x2 w2^T + bias``.

    Fused backward epilogues: ``ep_y`` + ``ep_fcoef`` (+ ``stats``) store the gradient masked by
    the ReLU of ``bn(ep_y)`` and emit the BatchNorm-backward partial sums per tile; ``ep_mask``
    zeroes the result where that tensor is <= 0; ``ep_y2`` + ``stats2`` (with ``ep_y`` and
    ``ep_mask``): partial sums of a second BatchNorm fed by the same gradient;
    ``ep_maskbits`` (uint8, one bit per output element, from :func:`bn_apply_fwd`) replaces
    ``ep_mask``."""
    B, Hin, Win, C, sb, sh, sw, xn = geom
    if LAUNCH_LOG is not None:
        LAUNCH_LOG.append(("conv_igemm", B, Hin, Win, C, Hout, Wout, n_out, R, S, ostride, idil, y.numel() // n_out,
                           int(addend is not None), int(stats is not None), int(ep_y is not None),
                           int(ep_mask is not None or ep_maskbits is not None), int(ep_y2 is not None),
                           int(bool(wt_rows)), c2))
    for t in (ep_y, ep_mask, ep_y2):
        if t is not None and t.numel() != y.numel():
            raise KernelError("conv_igemm: epilogue tensor does not match the output shape")
    if ep_fcoef is not None and ep_fcoef.__class__ is not int and \
            (ep_fcoef.numel() != 2 * n_out or not ep_fcoef.is_contiguous()):
        raise KernelError("conv_igemm: ep_fcoef must be a contiguous [2, N] tensor")
    if ep_y is not None and stats.__class__ is not int:
        need = conv_tiles_m_bwd(y.numel() // n_out, Hout, Wout, idil, ostride) * 2 * n_out
        if stats is None or stats.numel() < need:
            raise KernelError("conv_igemm: partial buffer too small for the fused reduce")
    if ep_y2 is not None and (stats2 is None or stats is None or stats2.numel() < stats.numel()):
        raise KernelError("conv_igemm: second partial buffer missing or too small")
    if ep_maskbits is not None and (ep_mask is not None or ep_maskbits.numel() * 8 < y.numel()):
        raise KernelError("conv_igemm: ep_maskbits excludes ep_mask and must cover the output")
    f = _fn("ddl_conv_igemm_ex4", "ppppp qq iiii qqq iiiiiiiiii pppppp ii pp qq i p p")
    check(f(x_ptr_tensor.data_ptr(), _p(w), y.data_ptr(), _p(addend), _p(stats), xn,
            w.numel() if w_numel is None else w_numel, B, Hin, Win, C, sb, sh, sw, Hout, Wout, n_out, R, S, ostride, pad, pad_w, idil,
            1 if addend_sub2 else 0, _p(ep_y), _p(ep_fcoef), _p(ep_mask), _p(ep_y2), _p(stats2),
            _p(ep_maskbits), 1 if wt_rows else 0, wt_rows, _p(x2), _p(w2),
            0 if x2 is None else x2.numel(), 0 if w2 is None else w2.numel(), c2, _p(bias),
            _raw_stream() if stream is None else stream),
          "conv_igemm")


@functools.lru_cache(maxsize=None)
def conv_tiles_m(M):
    return int(lib().ddl_conv_igemm_tiles_m(_i64(M)))


@functools.lru_cache(maxsize=None)
def gemm256_ok(M, N, K):
    """True when a plain-GEMM conv launch ([M][K] x [N][K]^T, no fused backward epilogue) is served by the
    256 x 256 / 8-wave kernel (csrc/kernels/gemm256.hip); the dispatch itself happens inside ``conv_igemm``."""
    return bool(lib().ddl_gemm256_ok(_i64(M), _i32(N), _i32(K)))


def gemm256_fwd(x2d, w2d, y2d, stats=None, stream=None):
    """y2d [M][N] = x2d [M][K] @ w2d [N][K]^T on the 256 x 256 / 8-wave kernel, whatever the dispatch policy says
    (N % 256 == 0, K % 32 == 0); ``stats`` ([ceil(M / 128)][2][N] fp32): sum | sum of squares of the rounded outputs."""
    M, Kd = x2d.shape
    N = w2d.shape[0]
    if tuple(y2d.shape) != (M, N) or w2d.shape[1] != Kd or not (x2d.is_contiguous() and w2d.is_contiguous() and y2d.is_contiguous()):
        raise KernelError("gemm256_fwd: dense [M][K], [N][K] and [M][N] operands expected")
    rows = 0 if stats is None else stats.shape[0]
    f = _fn("ddl_gemm256_fwd", "pppp qii qi p")
    check(f(x2d.data_ptr(), w2d.data_ptr(), y2d.data_ptr(), _p(stats), M, N, Kd, x2d.numel(), rows,
            _raw_stream() if stream is None else stream), "gemm256_fwd")


@functools.lru_cache(maxsize=None)
def conv_rows3x3_ok(C, N, R, S, stride, pad, H, W):
    """True when the row-streaming 3x3 kernel (csrc/kernels/conv_rows.hip) serves this layer, forward and
    backward-data; its statistics / BatchNorm-backward partial table has ONE row per image."""
    return bool(lib().ddl_conv_rows3x3_ok(_i32(C), _i32(N), _i32(R), _i32(S), _i32(stride), _i32(pad), _i32(H), _i32(W)))


def conv_rows3x3(x, w, y, stats=None, ep_y=None, ep_fcoef=None, stream=None, in_fcoef=None, z_out=None,
                 dy_y=None, dy_coef=None):
    """y = conv3x3(x, w), stride 1, pad 1, for dense NHWC [B, 64, H, 56] tensors and a [64][3][3][64] filter.
    ``stats`` ([B][2][64] fp32, tensor or raw address): per-image sum y | sum y^2; with ``ep_y`` + ``ep_fcoef``
    (backward-data of a layer behind BatchNorm + ReLU) y is the masked gradient g and the table holds
    sum g | sum g*ep_y.  ``in_fcoef`` ([2][64] fp32 scale | shift, tensor or raw address) + ``z_out`` (forward):
    ``x`` is the RAW input of a BatchNorm + ReLU in front of this conv; the kernel applies them to the rows it has
    staged, convolves the result and also writes it to ``z_out``.  ``dy_y`` + ``dy_coef`` ([3][64] fp32 ca | cb | cc,
    tensor or raw address; backward-data form only): ``x`` is the gradient with respect to a BatchNorm's output and
    ``dy_y`` that BatchNorm's input - the kernel convolves ca*x + cb*dy_y + cc, the backward apply pass folded in."""
    B, C, H, W = x.shape
    N = y.shape[1]
    if tuple(y.shape) != (B, N, H, W) or not conv_rows3x3_ok(C, N, 3, 3, 1, 1, H, W):
        raise KernelError("conv_rows3x3: unsupported shape %s -> %s" % (tuple(x.shape), tuple(y.shape)))
    for t in (x, y, ep_y, dy_y):
        if t is not None and (tuple(t.shape) != (B, C, H, W) or t.stride() != (H * W * C, 1, W * C, C)):
            raise KernelError("conv_rows3x3: tensors must be dense channels_last [B, 64, H, 56]")
    if w.numel() != N * 9 * C:
        raise KernelError("conv_rows3x3: filter must hold [64][3][3][64] elements")
    if stats is not None and stats.__class__ is not int and stats.numel() < B * 2 * N:
        raise KernelError("conv_rows3x3: statistics table needs one row per image")
    if LAUNCH_LOG is not None:
        LAUNCH_LOG.append(("conv_rows", B, H, W, C, H, W, N, 3, 3, 1, 1, B * H * W, int(ep_y is not None)))
    if z_out is not None and (tuple(z_out.shape) != (B, C, H, W) or z_out.stride() != x.stride()):
        raise KernelError("conv_rows3x3: z_out must look like the input")
    f = _fn("ddl_conv_rows3x3", "pppppp pp pp iiiii p")
    check(f(x.data_ptr(), w.data_ptr(), y.data_ptr(), _p(stats), _p(ep_y), _p(ep_fcoef), _p(in_fcoef), _p(z_out),
            _p(dy_y), _p(dy_coef), B, H, W, C, N, _raw_stream() if stream is None else stream), "conv_rows3x3")


@functools.lru_cache(maxsize=None)
def conv_tiles_m_bwd(M, Hout, Wout, idil, ostride):
    return int(lib().ddl_conv_igemm_tiles_m_bwd(_i64(M), _i32(Hout), _i32(Wout), _i32(idil),
                                                _i32(ostride)))


@functools.lru_cache(maxsize=None)
def conv_wgrad_splits(M, K, C, R, S, Hout=0, Wout=0, stride=1, B=0):
    """Reduction splits (= fp32 partial slabs) the weight-gradient launch of this problem writes.  ``Hout`` /
    ``Wout`` / ``stride`` / ``B`` let the library pick the row-streaming 3x3 kernel (it deals out whole images);
    a launch must be planned with the same values it is then called with."""
    return int(lib().ddl_conv_wgrad_splits_ex(_i64(M), _i32(K), _i32(C), _i32(R), _i32(S), _i32(Hout), _i32(Wout),
                                              _i32(stride), _i32(B)))


@functools.lru_cache(maxsize=None)
def conv_wgrad_colsum_ok(M, K, C):
    """True when the plain-GEMM weight gradient of this shape can emit the column sums of its x operand."""
    return bool(lib().ddl_conv_wgrad_colsum_ok(_i64(M), _i32(K), _i32(C)))


@functools.lru_cache(maxsize=None)
def conv_wgrad_rows_ok(K, C, R, S, Hout, Wout, stride, B):
    """True when the row-streaming weight-gradient kernel serves this (dense, < 2 GiB) problem."""
    return bool(lib().ddl_conv_wgrad_rows_ok(_i32(K), _i32(C), _i32(R), _i32(S), _i32(Hout), _i32(Wout), _i32(stride), _i32(B)))


def conv_wgrad(dy, x_ptr_tensor, geom, partial, K, R, S, stride, pad, pad_w, Hout, Wout, splits,
               stream=None, colsum=None, x_fcoef=None, dy_y=None, dy_coef=None):
    """``partial``: tensor or raw address; ``stream``: raw HIP stream (default: current).  ``colsum`` (raw address
    or tensor, plain GEMMs with :func:`conv_wgrad_colsum_ok`): receives [splits][2][C] fp32 column sums of x per
    split, both rows alike - a statistics-table layout that :func:`bn_colsum_fold` sums.  ``x_fcoef`` ([2][C] fp32
    scale | shift, raw address or tensor; only where :func:`conv_wgrad_rows_ok` and C = 64 on 56-pixel rows): ``x``
    is the RAW input of a BatchNorm + ReLU whose output is the conv's real input - applied on the staged rows.
    ``dy_y`` (tensor like ``dy``) + ``dy_coef`` ([3][K] fp32 ca | cb | cc; same kernel, K = 64): ``dy`` is the gradient
    with respect to a BatchNorm's output, ``dy_y`` that BatchNorm's input: the product uses ca*dy + cb*dy_y + cc."""
    B, Hin, Win, C, sb, sh, sw, xn = geom
    if _PROBE_SKIP_WGRAD == "all" or (_PROBE_SKIP_WGRAD == "late" and Hout <= 14) or (_PROBE_SKIP_WGRAD == "early" and Hout > 14):
        return   # timing probe only (gradients are garbage): what the step costs without these launches
    if LAUNCH_LOG is not None:
        LAUNCH_LOG.append(("conv_wgrad", B, Hin, Win, C, Hout, Wout, K, R, S, stride, 1, dy.numel() // K, splits))
    if dy_y is not None and (dy_y.shape != dy.shape or dy_y.stride() != dy.stride()):
        raise KernelError("conv_wgrad: dy_y must look like dy")
    f = _fn("ddl_conv_wgrad_ex3", "ppp qq iiii qqq iiiiiiiii pp pp p")
    check(f(dy.data_ptr(), x_ptr_tensor.data_ptr(), _p(partial), dy.numel(), xn, B, Hin, Win,
            C, sb, sh, sw, Hout, Wout, K, R, S, stride, pad, pad_w, splits, _p(colsum), _p(x_fcoef),
            _p(dy_y), _p(dy_coef),
            _raw_stream() if stream is None else stream), "conv_wgrad")


def splitk_reduce(partial, out, n, splits, slab, accumulate=False, stream=None, out_bf16=None):
    """``out``: tensor, or a raw address with ``out_bf16`` saying what it holds."""
    if _PROBE_SKIP_WGRAD == "reduce":
        return   # timing probe only
    f = _fn("ddl_splitk_reduce", "pp qiq ii p")
    if out_bf16 is None:
        out_bf16 = out.dtype == torch.bfloat16
    check(f(_p(partial), _p(out), n, splits, slab,
            1 if out_bf16 else 0, 1 if accumulate else 0,
            _raw_stream() if stream is None else stream), "splitk_reduce")


def splitk_reduce_lin(partial, out, n, splits, slab, accumulate, coef, C, T, colz, row_len, mean, stream=None):
    """out (+)= ca[c]*sum_s partial[s] + cb[c]*T + (cc[c] + cb[c]*mean[c])*colz[j] (BatchNorm backward by
    linearity; ``T`` = W times the CENTRED Gram matrix from :func:`splitk_reduce_centered`)."""
    f = _fn("ddl_splitk_reduce_lin", "pp qiq ii p i pp i p p")
    check(f(_p(partial), out.data_ptr(), n, splits, slab, 1 if out.dtype == torch.bfloat16 else 0,
            1 if accumulate else 0, _p(coef), C, _p(T), _p(colz), row_len, _p(mean),
            _raw_stream() if stream is None else stream), "splitk_reduce_lin")


@functools.lru_cache(maxsize=None)
def lin_fused_wgs(M, C, C2):
    """Workgroups (= slabs of the raw product, = statistics rows) of :func:`lin_fused` for M pixels, or 0 when the
    fused kernel does not serve the shape (csrc/kernels/lin_fused.hip: 256 wide / 64 narrow channels)."""
    return int(lib().ddl_lin_fused_wgs(_i64(M), _i32(C), _i32(C2)))


@functools.lru_cache(maxsize=None)
def lin_fused_gram_ok(C, C2):
    """True when :func:`lin_fused` of this shape can also accumulate z2^T z2 and the column sums of z2."""
    return bool(lib().ddl_lin_fused_gram_ok(_i32(C), _i32(C2)))


def lin_fused(g, z2, b1, q, bias, y_prev, fcoef, g2, stats, partial, M, C, C2, nwg, stream=None, gram=None, colsum=None):
    """One pass over g [M][C] and z2 [M][C2]: g2 = mask(g b1^T + z2 q^T + bias) (masked by relu'(bn(y_prev)), with
    the sums sum g2 | sum g2*y_prev in ``stats`` [nwg][2][C2]) and the slabs of g^T z2 in ``partial`` [nwg][C][C2];
    with ``gram`` [nwg][C2][C2] + ``colsum`` [nwg][2][C2] (where :func:`lin_fused_gram_ok`) also the slabs of z2^T z2
    and the column sums of z2 (both rows of a slab's table alike)."""
    f = _fn("ddl_lin_fused", "pppppppppp pp qiii p")
    check(f(_p(g), _p(z2), _p(b1), _p(q), _p(bias), _p(y_prev), _p(fcoef), _p(g2), _p(stats), _p(partial),
            _p(gram), _p(colsum), M, C, C2, nwg, _raw_stream() if stream is None else stream), "lin_fused")


def splitk_reduce_centered(partial, out, C, splits, colz, M, stream=None):
    """out (bf16 [C][C]) = sum_s partial[s] - colz (x) colz / M (centred in fp32, rounded once)."""
    f = _fn("ddl_splitk_reduce_centered", "pp ii p q p")
    check(f(_p(partial), _p(out), C, splits, _p(colz), M, _raw_stream() if stream is None else stream),
          "splitk_reduce_centered")


def bn_lin_prep(wt, ld, coef, C, rows, b1, a2, bias):
    f = _fn("ddl_bn_lin_prep", "pi pii ppp p")
    check(f(_p(wt), ld, _p(coef), C, rows, _p(b1), _p(a2), _p(bias), _raw_stream()), "bn_lin_prep")


def bn_lin_prep_q(wt, ld, coef, C, rows, b1, q, bias):
    """:func:`bn_lin_prep` and Q = a2 wt^T (bf16 [rows][rows]) in one launch (a2 is never stored)."""
    f = _fn("ddl_bn_lin_prep_q", "pi pii ppp p")
    check(f(_p(wt), ld, _p(coef), C, rows, _p(b1), _p(q), _p(bias), _raw_stream()), "bn_lin_prep_q")


def bn_colsum_fold(partial, T, C, out, stream=None):
    """out[2][C] = column sums of a [T][2][C] partial table (fixed order)."""
    f = _fn("ddl_bn_colsum_fold", "p ii p p")
    check(f(_p(partial), T, C, _p(out), _raw_stream() if stream is None else stream), "bn_colsum_fold")


# ---- batch norm / pooling / loss (csrc/kernels/nn_kernels.hip) -------------------------------
def bn_stats_partial(y2d, partial, M, C, rows_per_tile, stream=None):
    """``partial``: tensor or raw address; ``stream``: raw HIP stream (default: current)."""
    f = _fn("ddl_bn_stats_partial", "pp qii p")
    check(f(y2d.data_ptr(), _p(partial), M, C, rows_per_tile, _raw_stream() if stream is None else stream),
          "bn_stats_partial")


@functools.lru_cache(maxsize=None)
def _fold_slices(T):
    return int(lib().ddl_bn_fold_slices(_i32(T)))


@functools.lru_cache(maxsize=None)
def bn_direct_max():
    """Longest statistics / BatchNorm-backward partial table the finalize kernels sum in one launch."""
    return int(lib().ddl_bn_direct_max())


def bn_fwd_finalize(partial, T, C, M, rows_per_tile, gamma, beta, rmean, rvar, momentum, eps,
                    mean, invstd, scale, shift, work=None):
    """``work``: [fold_slices(T), 2, C] float64 scratch, needed (and allocated here unless given)
    only for tables of more than 512 tiles."""
    if work is None and T > bn_direct_max():
        work = torch.empty((_fold_slices(T), 2, C), dtype=torch.float64, device=partial.device)
    f = _fn("ddl_bn_fwd_finalize", "p iiqi pppp ff ppppp p")
    check(f(_p(partial), T, C, M, rows_per_tile, gamma.data_ptr(), beta.data_ptr(), _p(rmean),
            _p(rvar), momentum, eps, _p(mean), _p(invstd), _p(scale), _p(shift), _p(work),
            _raw_stream()), "bn_fwd_finalize")


def bn_eval_coeffs(gamma, beta, rmean, rvar, eps, scale, shift):
    check(lib().ddl_bn_eval_coeffs(ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), _f32(eps),
                                   _i32(gamma.numel()), ptr(scale), ptr(shift), stream()),
          "bn_eval_coeffs")


def bn_apply_fwd(y, scale, shift, res, z, M, C, relu, res_scale=None, res_shift=None, bits=None):
    """z = act(y*scale + shift [+ res]); with ``res_scale``/``res_shift`` the residual is the raw
    output of the projection conv and its BatchNorm is applied on the fly.  ``bits`` (uint8,
    M*C/8 bytes): also store ``z > 0`` as one bit per element."""
    if bits is not None and bits.numel() * 8 < M * C:
        raise KernelError("bn_apply_fwd: mask buffer too small")
    f = _fn("ddl_bn_apply_fwd_ex2", "pppppppp qii p")
    check(f(y.data_ptr(), _p(scale), _p(shift), _p(res), _p(res_scale), _p(res_shift),
            z.data_ptr(), _p(bits), M, C, 1 if relu else 0, _raw_stream()), "bn_apply_fwd")


@functools.lru_cache(maxsize=None)
def bn_bwd_blocks(M):
    return int(lib().ddl_bn_bwd_blocks(_i64(M)))


def bn_bwd_reduce(dz, y, z, fcoef, partial, M, C, mask, G):
    f = _fn("ddl_bn_bwd_reduce", "ppppp qiii p")
    check(f(dz.data_ptr(), y.data_ptr(), _p(z), _p(fcoef), _p(partial), M, C, mask, G,
            _raw_stream()), "bn_bwd_reduce")


def bn_bwd_finalize(partial, G, C, M, gamma, mean, invstd, dgamma, dbeta, coef, accumulate=False,
                    work=None):
    """``work``: [fold_slices(G), 2, C] float32 scratch for tables of more than 512 rows."""
    if work is None and G > bn_direct_max():
        work = torch.empty((_fold_slices(G), 2, C), dtype=torch.float32, device=gamma.device)
    f = _fn("ddl_bn_bwd_finalize", "p iiq pppppp i pp")
    check(f(_p(partial), G, C, M, gamma.data_ptr(), _p(mean), _p(invstd), _p(dgamma), _p(dbeta),
            _p(coef), 1 if accumulate else 0, _p(work), _raw_stream()), "bn_bwd_finalize")


def bn_bwd_apply(dz, y, z, fcoef, coef, dy, gout, M, C, mask):
    f = _fn("ddl_bn_bwd_apply", "ppppppp qii p")
    check(f(dz.data_ptr(), y.data_ptr(), _p(z), _p(fcoef), _p(coef), dy.data_ptr(), _p(gout), M,
            C, mask, _raw_stream()), "bn_bwd_apply")


def maxpool_fwd(x, out, idx, B, H, W, C):
    check(lib().ddl_maxpool3x3s2_fwd(ptr(x), ptr(out), ptr(idx), _i32(B), _i32(H), _i32(W), _i32(C),
                                     stream()), "maxpool_fwd")


def bn_relu_maxpool_fwd(x, scale, shift, out, idx, B, H, W, C):
    """out = maxpool3x3s2(relu(x*scale + shift)) without materialising the activation."""
    f = _fn("ddl_bn_relu_maxpool3x3s2_fwd", "ppppp iiii p")
    check(f(x.data_ptr(), _p(scale), _p(shift), out.data_ptr(), idx.data_ptr(), B, H, W, C,
            _raw_stream()), "bn_relu_maxpool_fwd")


def stem_pool_bn_bwd(dpool, idx, y, fcoef, coef, partial, dy, B, H, W, C, G, apply):
    """Fused stem backward pass over y [B,H,W,C]: reduce partials (apply=False) or dy (apply=True)."""
    f = _fn("ddl_stem_pool_bn_bwd", "ppppppp iiiiii p")
    check(f(dpool.data_ptr(), idx.data_ptr(), y.data_ptr(), _p(fcoef), _p(coef), _p(partial), _p(dy),
            B, H, W, C, G, 1 if apply else 0, _raw_stream()), "stem_pool_bn_bwd")


def maxpool_bwd(dy, idx, dx, B, H, W, C):
    check(lib().ddl_maxpool3x3s2_bwd(ptr(dy), ptr(idx), ptr(dx), _i32(B), _i32(H), _i32(W), _i32(C),
                                     stream()), "maxpool_bwd")


def avgpool_fwd(x, out, B, HW, C):
    check(lib().ddl_avgpool_fwd(ptr(x), ptr(out), _i32(B), _i32(HW), _i32(C), stream()), "avgpool_fwd")


def avgpool_bwd(dy, dx, B, HW, C):
    check(lib().ddl_avgpool_bwd(ptr(dy), ptr(dx), _i32(B), _i32(HW), _i32(C), stream()), "avgpool_bwd")


def softmax_xent(logits, labels, loss_rows, dlogits, B, N, ldl, ldd, grad_scale, go=None, loss_mean=None,
                 ticket=None):
    """K13.  ``loss_rows`` and/or ``dlogits`` may be None (loss only / gradient only); ``go``: device
    scalar multiplied into the gradient; ``loss_mean`` + ``ticket``: the last block stores the mean
    of the row losses (fixed order) - no separate reduction launch."""
    f = _fn("ddl_softmax_xent_ex", "pppp iiii f ppp p")
    check(f(logits.data_ptr(), labels.data_ptr(), _p(loss_rows), _p(dlogits), B, N, ldl, ldd, grad_scale,
            _p(go), _p(loss_mean), _p(ticket), _raw_stream()), "softmax_xent")


def label_rank(logits, labels):
    """K21: per-row rank (int32) of the label's logit in top-k order; top-k hit = rank < k."""
    B, N = logits.shape
    assert logits.dtype == torch.bfloat16 and logits.stride(1) == 1 and labels.dtype == torch.int64
    rank = torch.empty(B, dtype=torch.int32, device=logits.device)
    f = _fn("ddl_label_rank", "ppp iii p")
    check(f(logits.data_ptr(), labels.data_ptr(), rank.data_ptr(), B, N, logits.stride(0),
            _raw_stream()), "label_rank")
    return rank


def bias_add_(y, bias, M, N):
    check(lib().ddl_bias_add(ptr(y), ptr(bias), _i64(M), _i32(N), stream()), "bias_add")


def colsum(a, out, M, N, ld, accumulate=False):
    f = _fn("ddl_colsum_ex", "pp iiii p")
    check(f(a.data_ptr(), out.data_ptr(), M, N, ld, 1 if accumulate else 0, _raw_stream()), "colsum")


# ---- step-side helpers (csrc/kernels/step_aux.hip): no framework dispatch inside the training step
def pool_gather(pool, pool_labels, out, out_labels, state, batch, seed):
    """Synthetic loader: draw ``batch`` pool rows (Philox of ``seed`` and the device-side batch
    counter ``state[0]``, which the kernel advances) and gather images + labels, one launch."""
    row_bytes = pool[0].numel() * pool.element_size()
    f = _fn("ddl_pool_gather", "ppppp ii q Q p")
    check(f(pool.data_ptr(), pool_labels.data_ptr(), out.data_ptr(), out_labels.data_ptr(), state.data_ptr(),
            pool.shape[0], batch, row_bytes, seed & 0xFFFFFFFFFFFFFFFF, _raw_stream()), "pool_gather")


def stem_weight_pack(w, wp):
    f = _fn("ddl_stem_weight_pack", "pp ii p")
    check(f(w.data_ptr(), wp.data_ptr(), w.shape[0], w.shape[1], _raw_stream()), "stem_weight_pack")


def stem_dw_unpack(dwp, dw, K, C, accumulate):
    f = _fn("ddl_stem_dw_unpack", "pp iiii p")
    check(f(_p(dwp), dw.data_ptr(), K, C, 1 if dw.dtype == torch.bfloat16 else 0, 1 if accumulate else 0,
            _raw_stream()), "stem_dw_unpack")


def bump_counters(table, n):
    f = _fn("ddl_bump_counters", "p i p")
    check(f(table.data_ptr(), n, _raw_stream()), "bump_counters")


def memset_async(t, value=0):
    f = _fn("ddl_memset_async", "p i q p")
    check(f(t.data_ptr(), value, t.numel() * t.element_size(), _raw_stream()), "memset_async")


def weight_flip_transpose(base_src_addr, base_dst_addr, table, nentries, total_blocks):
    check(lib().ddl_weight_flip_transpose(ctypes.c_void_p(base_src_addr), ctypes.c_void_p(base_dst_addr),
                                          ptr(table), _i32(nentries), _i32(total_blocks), stream()),
          "weight_flip_transpose")


def stem_pack(x, out):
    B, C, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    check(lib().ddl_stem_pack(ptr(x), ptr(out), _i32(B), _i32(H), _i32(W), _i32(C), _i64(sb), _i64(sh),
                              _i64(sw), _i64(sc), stream()), "stem_pack")


def subsample2(x):
    """Every second row and column of a dense NHWC (channels_last) bf16 tensor - what a stride-2 1x1
    projection reads - as a dense NHWC tensor (replaces ``x[:, :, ::2, ::2].contiguous()``)."""
    B, C, H, W = x.shape
    if x.dtype != torch.bfloat16 or x.stride() != (H * W * C, 1, W * C, C):
        raise ValueError("subsample2: dense channels_last bf16 input expected, got %s %s" % (x.dtype, x.stride()))
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    out = torch.empty_strided((B, C, Ho, Wo), (Ho * Wo * C, 1, Wo * C, C), dtype=x.dtype, device=x.device)
    check(lib().ddl_subsample2(ptr(x), ptr(out), _i32(B), _i32(H), _i32(W), _i32(C), stream()), "subsample2")
    return out


# ---- direct xGMI all-reduce and multi-tensor pack (csrc/kernels/xgmi_allreduce.hip) ----------
def xgmi_allreduce(buf_ptrs, flag_ptrs, out_ptrs, count, rank, nranks, is_bf16, one_shot, epoch,
                   scale, timeout_word, blocks=16, spin_limit=2_000_000, simulate=False):
    """Raw launcher: ``buf_ptrs`` / ``flag_ptrs`` / ``out_ptrs`` are lists of device addresses
    indexed by rank (``out_ptrs`` only for one-shot).  ``simulate`` plays every rank in one
    launch (single-device protocol test)."""
    arr = ctypes.c_void_p * nranks
    outs = arr(*[o or 0 for o in out_ptrs]) if out_ptrs else None
    check(lib().ddl_xgmi_allreduce(arr(*buf_ptrs), arr(*flag_ptrs), outs, _i64(count),
                                   _i32(rank), _i32(nranks), _i32(bool(is_bf16)),
                                   _i32(bool(one_shot)), ctypes.c_uint(epoch), _f32(scale),
                                   _i64(spin_limit), ptr(timeout_word), _i32(blocks),
                                   _i32(bool(simulate)), stream()), "xgmi_allreduce")


def pack_scale_cast(tensors, flat, offsets, scale=1.0, unpack=False):
    """K17: gather ``tensors`` (all fp32 or all bf16, contiguous) into ``flat`` at element
    ``offsets`` with scaling and dtype cast — or scatter back when ``unpack``."""
    import numpy as np

    src_dt = tensors[0].dtype
    assert all(t.dtype == src_dt and t.is_contiguous() for t in tensors)
    tab = np.array([(t.data_ptr(), o, t.numel()) for t, o in zip(tensors, offsets)], dtype=np.int64)
    table = torch.from_numpy(tab).to(flat.device)
    check(lib().ddl_pack_scale_cast(ptr(table), _i32(len(tensors)), ptr(flat),
                                    _i32(src_dt == torch.bfloat16),
                                    _i32(flat.dtype == torch.bfloat16), _f32(scale),
                                    _i32(bool(unpack)), stream()), "pack_scale_cast")
    return table  # keep alive until the stream has consumed it
