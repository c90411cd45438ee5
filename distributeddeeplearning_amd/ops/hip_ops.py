"""autograd wrappers around the HIP kernels (GPU path of ``ops.functional``).

Every op here runs on the hand-written gfx950 kernels of ``csrc/kernels`` (SURVEY §2.6
K1–K13): implicit-GEMM convolution forward / backward-data / backward-weights on bf16 MFMA,
BatchNorm statistics fused into the conv epilogue, fused BN-apply(+residual)+ReLU forward and
backward, pooling, the dense head and softmax cross-entropy.  Tensors are logically NCHW and
physically NHWC (``torch.channels_last``) bf16; BatchNorm parameters and statistics are fp32.

Replaces what the reference gets from cuDNN/ATen through torchvision's ResNet-50
(``HorovodPytorch/src/imagenet_pytorch_horovod.py:214-217,323,340``).
"""
from __future__ import annotations

import numpy as np
import torch
from torch.autograd import Function

from . import kernels as K

IMPLEMENTED = {"conv2d", "batchnorm_act", "maxpool3x3s2", "global_avgpool", "linear",
               "softmax_cross_entropy"}

_BF16 = torch.bfloat16
_CL = torch.channels_last


def _empty_nhwc(B, C, H, W, device, dtype=_BF16):
    return torch.empty((B, C, H, W), dtype=dtype, device=device, memory_format=_CL)


def _as_nhwc(t):
    if t.dim() == 4 and (t.stride(1) == 1 or t.shape[1] == 1) and \
            t.is_contiguous(memory_format=_CL):
        return t
    return t.contiguous(memory_format=_CL)


def _geom(x):
    B, C, H, W, sb, sh, sw = K.nhwc_dims(x)
    return (B, H, W, C, sb, sh, sw, K.storage_numel_from(x))


def _krsc_ok(w):
    """conv weight [K, C, R, S] whose memory is dense [K][R][S][C]."""
    if w.dim() == 2:
        return w.is_contiguous()
    return w.permute(0, 2, 3, 1).is_contiguous()


# ----------------------------------------------------------------------------------------------
# flipped / transposed weights for backward-data: dst[c][R-1-r][S-1-s][k] = w[k][r][s][c]
# All registered weights are refreshed by ONE kernel launch per optimizer step.
# ----------------------------------------------------------------------------------------------
class _WeightTransposes:
    _DT = np.dtype([("src", "<i8"), ("dst", "<i8"), ("K", "<i4"), ("R", "<i4"), ("S", "<i4"),
                    ("C", "<i4"), ("Kpad", "<i4"), ("block0", "<i4")])

    def __init__(self):
        self.reset()

    def reset(self):
        self.entries = {}
        self.dirty = True
        self.epoch = 0
        self.done_epoch = -1
        self.table = None
        self.dst = None
        self.total_blocks = 0
        self.base_src = 0
        self.pending = None     # event of a refresh running on the weight-gradient stream
        self.captured_epoch = -1

    def bump(self):
        self.epoch += 1

    def _register(self, w):
        if w.dim() == 4:
            Kc, C, R, S = w.shape
        else:
            (Kc, C), R, S = w.shape, 1, 1
        kpad = (Kc + 31) // 32 * 32
        self.entries[id(w)] = {"w": w, "ptr": w.data_ptr(), "K": Kc, "C": C, "R": R, "S": S,
                               "Kpad": kpad, "version": -1, "dst_off": 0}
        self.dirty = True

    def _rebuild(self, device):
        assert K.lib().ddl_wt_entry_bytes() == self._DT.itemsize
        # drop entries whose tensor storage moved or died
        live = {k: e for k, e in self.entries.items() if e["w"].data_ptr() == e["ptr"]}
        self.entries = live
        ents = list(live.values())
        self.base_src = min(e["ptr"] for e in ents)
        tab = np.zeros(len(ents), dtype=self._DT)
        off = blocks = 0
        for i, e in enumerate(ents):
            e["dst_off"] = off
            nb = ((e["Kpad"] + 63) // 64) * ((e["C"] + 63) // 64) * e["R"] * e["S"]   # 64 x 64 tiles per tap
            if e["C"] % 8 or (e["ptr"] - self.base_src) % 16:
                raise K.KernelError("weight transposition needs 16-byte aligned weights with channels % 8 == 0")
            tab[i] = ((e["ptr"] - self.base_src) // 2, off, e["K"], e["R"], e["S"], e["C"],
                      e["Kpad"], blocks)
            blocks += nb
            off += (e["C"] * e["R"] * e["S"] * e["Kpad"] + 7) // 8 * 8
        self.total_blocks = blocks
        self.dst = torch.empty(off, dtype=_BF16, device=device)
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(device)
        self.n = len(ents)
        self.dirty = False

    def _launch(self):
        K.weight_flip_transpose(self.base_src, self.dst.data_ptr(), self.table, self.n, self.total_blocks)
        for x in self.entries.values():
            x["version"] = x["w"]._version
        self.done_epoch = self.epoch

    def refresh_async(self):
        """Called right after an optimizer step: transpose on the weight-gradient stream, which is
        idle during the forward pass, instead of in front of the first backward-data kernel (the
        launch sat on the critical path of backward: ~0.13 ms per ResNet-50 step)."""
        if self.dirty or self.table is None or not _SIDE["on"] or _WT_KRSC:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        # raw stream handles and events (no torch stream switch): the calls are recorded by the launch
        # tape like every kernel launch, and cost a third of the torch objects on the host
        side = _side_stream(self.dst.device).cuda_stream
        K.stream_wait_stream(side, K._raw_stream(), _SIDE["event"])   # after the optimizer's writes
        with K.launch_on(side):
            self._launch()
        self.pending = side

    def get(self, w):
        e = self.entries.get(id(w))
        if e is None or e["ptr"] != w.data_ptr():
            self._register(w)
            e = self.entries[id(w)]
        if self.pending is not None:   # first use after an asynchronous refresh: order this stream behind it
            K.stream_wait_stream(K._raw_stream(), self.pending, _SIDE["event_wt"])
            self.pending = None
        stale = self.dirty or self.done_epoch != self.epoch or e["version"] != w._version
        if not stale and self.captured_epoch != self.epoch and torch.cuda.is_current_stream_capturing():
            # a step being captured into a HIP graph must contain its own refresh: the replays never
            # come through here, and the asynchronous refresh above ran outside the graph
            stale = True
            self.captured_epoch = self.epoch
        if self.dirty:
            self._rebuild(w.device)
        if stale:
            self._launch()
        n = e["C"] * e["R"] * e["S"] * e["Kpad"]
        return self.dst[e["dst_off"]: e["dst_off"] + n]


_WT = _WeightTransposes()
import os as _os0
# DDL_DGRAD_KRSC=1: backward-data reads the FORWARD filter and transposes it with ds_read_b64_tr_b16 on
# the way from LDS to the matrix cores (no transposed copy, no transposition kernel).  Measured 0.4 ms
# per ResNet-50 step SLOWER than the copy (twice the LDS read instructions for the weight operand at
# 2-3 waves per SIMD, profiles/r2_notes.md), so the default keeps the copy, refreshed once per step
# on the weight-gradient stream during the forward pass.
_WT_KRSC = _os0.environ.get("DDL_DGRAD_KRSC", "0") == "1"


def _dgrad_w(w):
    """(weight operand, wt_rows) of a backward-data launch."""
    if _WT_KRSC:
        return w, w.shape[0]
    return _WT.get(w), 0


def weights_changed():
    """Tell the transposed-weight cache that parameters were updated outside autograd's view
    (the fused optimizers write them through raw pointers)."""
    _WT.bump()
    _WT.refresh_async()


def reset_caches():
    _WT.reset()


# ----------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------
def _is_stem(w, stride, padding):
    return w.dim() == 4 and w.shape[1] <= 4 and w.shape[2] == 7 and w.shape[3] == 7 and \
        stride == 2 and padding == 3


def _wgrad_into(dy, x_ptr_tensor, geom, w_like, Kc, R, S, stride, pad, pad_w, Ho, Wo, n_valid=None,
                accumulate=False, stream=None, workspace=None, x_fcoef=None, dy_y=None, dy_coef=None):
    """Backward-weights: split-K partials then ordered reduction into a tensor shaped like
    ``w_like`` (memory [K][R][S][C]).  ``stream`` (raw handle) / ``workspace`` (callable: bytes ->
    device address valid on that stream): the side-stream path, which neither switches torch's
    current stream nor allocates."""
    B, Hin, Win, C = geom[0], geom[1], geom[2], geom[3]
    M = B * Ho * Wo
    splits = K.conv_wgrad_splits(M, Kc, C, R, S, Ho, Wo, stride, B)
    slab = Kc * R * S * C
    if workspace is None:
        partial = torch.empty((splits, slab), dtype=torch.float32, device=dy.device)
    else:
        partial = workspace(4 * splits * slab)
    K.conv_wgrad(dy, x_ptr_tensor, geom, partial, Kc, R, S, stride, pad, pad_w, Ho, Wo, splits,
                 stream=stream, x_fcoef=x_fcoef, dy_y=dy_y, dy_coef=dy_coef)
    n = slab if n_valid is None else n_valid
    K.splitk_reduce(partial, w_like, n, splits, slab, accumulate, stream=stream)
    return w_like


# ---- gradient sinks --------------------------------------------------------------------------
# A fused block writes weight gradients straight into ``param.grad`` (the flat gradient arena of
# parallel.bucket.GradArena: wgrad output lands in the all-reduce buffer, no extra add pass) and
# returns None to autograd for them; listeners (DistributedOptimizer) are told which parameter
# just became ready, which is what autograd's post-accumulate hook would have done.
_grad_listeners = []
_listener_params = {}   # id(listener) -> set of parameter ids it serves


def add_grad_listener(fn):
    if fn not in _grad_listeners:
        _grad_listeners.append(fn)


def remove_grad_listener(fn):
    if fn in _grad_listeners:
        _grad_listeners.remove(fn)
    _listener_params.pop(id(fn), None)


def claim_grad_listener(fn, params):
    """Register ``fn`` as THE listener for ``params`` (an iterable of ``id(parameter)``): an older
    listener serving any of them (an optimizer that was re-wrapped or rebuilt over the same model)
    is dropped - two optimizers must never both exchange one parameter's gradient."""
    mine = set(params)
    for old in list(_grad_listeners):
        if _listener_params.get(id(old), set()) & mine:
            remove_grad_listener(old)
    add_grad_listener(fn)
    _listener_params[id(fn)] = mine


def _direct_grad_ok(p):
    g = p.grad
    return g is not None and g.dtype in (torch.float32, _BF16) and g.shape == p.shape and \
        g.stride() == p.stride() and g.data_ptr() % 16 == 0


# Inside a fused block's backward, readiness notifications are held back until the next kernel of
# the dependent chain has been issued: a notification may launch a gradient collective (~50-100 us
# of host time), and in the late layers the host is what the main stream waits for.
_DEFER = {"on": False, "items": []}


def _tell(p):
    for fn in list(_grad_listeners):
        if fn(p) is False:       # a listener whose optimizer is gone
            remove_grad_listener(fn)


def _notify(p):
    if _DEFER["on"]:
        _DEFER["items"].append(p)
        return
    _tell(p)


def _flush_notifications():
    items, _DEFER["items"] = _DEFER["items"], []
    for p in items:
        _tell(p)


# ---- raw (autograd-free) building blocks -------------------------------------------------------
import os as _os_rows


def _rows_ok(x, w, stride, padding):
    """The row-streaming 3x3 kernel serves this conv (dense NHWC input, dense [K][3][3][C] filter)."""
    if w.dim() != 4 or x.dim() != 4:
        return False
    B, C, H, W = x.shape
    Kc, _, R, S = w.shape
    return K.conv_rows3x3_ok(C, Kc, R, S, stride, padding, H, W) and \
        x.stride() == (H * W * C, 1, W * C, C) and x.numel() * 2 < (1 << 31) - 1


# BatchNorm apply + ReLU of a bottleneck's FIRST unit inside the row-streaming kernels that consume it (the 3x3 conv in
# forward, that conv's weight gradient in backward): DDL_CONV_ROWS_BN_IN = 2 (default) the activated tensor is never
# written - its only two readers transform the raw conv output on the rows they stage in LDS; 1 = the forward conv
# transforms and also writes the tensor out for a plain weight gradient (bandwidth-bound: breaks even); 0 = separate
# apply pass.  16.75 -> 16.69 ms per step, bit-identical training (profiles/r3_notes.md section 19).
_ROWS_BN_IN = int(_os_rows.environ.get("DDL_CONV_ROWS_BN_IN", "2"))


# BatchNorm-backward APPLY of a 3x3 unit inside the two row-streaming kernels that consume its result (the conv's
# backward-data and weight-gradient kernels): dy = ca*g + cb*y + cc is formed on the rows they stage in LDS and the
# materialised dy - a read of two tensors and a write of one on the dependent chain - never exists.  Bit-identical,
# one tensor pass of eight less, the dependent chain 56 us per block shorter - and the step is SLOWER with the
# weight-gradient stream running (16.64 -> 16.72 ms; ResNet-34 9.53 -> 9.58): the two row-streaming kernels own a
# whole CU each (LDS) and take turns, the separate apply pass rode along with either for free, and the backward-data
# kernel with a third input ring takes 105 instead of 78 us in the step.  Without the side stream the folded form
# wins (18.80 -> 18.72 ms).  Hence OFF by default; DDL_CONV_ROWS_BN_BWD=1 switches it on
# (profiles/r3_notes.md section 20).
_ROWS_BN_BWD = int(_os_rows.environ.get("DDL_CONV_ROWS_BN_BWD", "0"))


def _rows_bn_bwd_ok(g, w, stride, padding, inp, y_prev):
    """Both consumers of this unit's BatchNorm-backward output are row-streaming kernels that can form it
    themselves: ``g`` the (masked) gradient w.r.t. the BatchNorm's output, ``w`` the unit's conv, ``inp`` that
    conv's input (or the raw tensor it is derived from), ``y_prev`` the tensor behind the backward-data epilogue."""
    if not _ROWS_BN_BWD or _WT_KRSC or w.dim() != 4:
        return False
    Kc, C, R, S = w.shape
    B, _, H, W = g.shape
    return Kc == 64 and C == 64 and W == 56 and _rows_ok(g, w, stride, padding) and \
        K.conv_wgrad_rows_ok(Kc, C, R, S, H, W, stride, B) and tuple(inp.shape) == tuple(g.shape) and \
        inp.stride() == g.stride() and y_prev.stride() == g.stride() and tuple(y_prev.shape) == tuple(g.shape)


def _rows_bn_in_ok(x, w, stride, padding, w_next, stride_next, pad_next):
    """Unit (x, w) is a conv whose BatchNorm + ReLU output only feeds the next conv (w_next), and that conv is
    served by the row-streaming kernel on this unit's output shape."""
    if w.dim() != 4 or w_next.dim() != 4:
        return False
    B, C, H, W = x.shape
    Kc, _, R, S = w.shape
    Ho = (H + 2 * padding - R) // stride + 1
    Wo = (W + 2 * padding - S) // stride + 1
    Kn, Cn, Rn, Sn = w_next.shape
    return Cn == Kc and K.conv_rows3x3_ok(Kc, Kn, Rn, Sn, stride_next, pad_next, Ho, Wo) and \
        B * Ho * Wo * Kc * 2 < (1 << 31) - 1


def _conv_fwd(x, w, stride, padding, want_stats):
    B, C, H, W = x.shape
    Kc, Cw, R, S = w.shape
    if Cw != C:
        raise K.KernelError("conv: input has %d channels, weight expects %d" % (C, Cw))
    Ho = (H + 2 * padding - R) // stride + 1
    Wo = (W + 2 * padding - S) // stride + 1
    y = _empty_nhwc(B, Kc, Ho, Wo, x.device)
    stats = None
    if _rows_ok(x, w, stride, padding):
        if want_stats:
            stats = torch.empty((B, 2, Kc), dtype=torch.float32, device=x.device)
        K.conv_rows3x3(x, w, y, stats=stats)
        return y, stats
    if want_stats:
        stats = torch.empty((K.conv_tiles_m(B * Ho * Wo), 2, Kc), dtype=torch.float32,
                            device=x.device)
    K.conv_igemm(x, _geom(x), w, y, Kc, R, S, stride, padding, padding, 1, Ho, Wo, stats=stats)
    return y, stats


def _conv_dgrad(dy, w, x_shape, stride, padding, addend=None, addend_sub2=False, mask_src=None,
                bn_y=None, bn_y2=None, mask_bits=None):
    """dx of a convolution; ``addend`` is summed in the epilogue (shortcut gradient) and
    ``mask_src`` (the ReLU output that IS the conv input) zeroes dx where it is <= 0.  With
    ``bn_y`` (the pre-BatchNorm tensor behind that ReLU; needs ``mask_src``) the kernel also
    emits that BatchNorm's backward partial sums and ``(dx, partial)`` is returned; ``bn_y2``
    (a second BatchNorm input fed by the same gradient: the projection shortcut's conv output)
    adds a table of its own: ``(dx, partial, partial2)``.  ``mask_bits`` (uint8, one bit per
    element of ``mask_src > 0``, written by the forward BatchNorm apply) is read instead of
    ``mask_src`` when given."""
    B, C, H, W = x_shape
    _, _, R, S = w.shape
    wt, wrows = _dgrad_w(w)
    if R == 1 and S == 1 and stride == 2 and padding == 0 and addend is None and mask_src is None:
        # 1x1 stride 2: only even pixels receive a gradient; compute them as a plain GEMM and
        # scatter (the fused block never gets here: it hands the compact result to the next
        # backward-data kernel as a sub-sampled addend)
        dx = torch.zeros((B, C, H, W), dtype=_BF16, device=dy.device).contiguous(memory_format=_CL)
        dx[:, :, ::2, ::2] = _conv_dgrad_1x1s2_compact(dy, w)
        return dx
    dx = _empty_nhwc(B, C, H, W, dy.device)
    partial = partial2 = None
    if bn_y is not None:
        T = K.conv_tiles_m_bwd(B * H * W, H, W, stride, 1)
        partial = torch.empty((T, 2, C), dtype=torch.float32, device=dy.device)
        if bn_y2 is not None:
            partial2 = torch.empty((T, 2, C), dtype=torch.float32, device=dy.device)
    K.conv_igemm(dy, _geom(dy), wt, dx, C, R, S, 1, R - 1 - padding, S - 1 - padding, stride, H, W,
                 addend=addend, addend_sub2=addend_sub2,
                 ep_mask=mask_src if mask_bits is None else None, ep_y=bn_y,
                 stats=partial, ep_y2=bn_y2 if bn_y is not None else None, stats2=partial2,
                 ep_maskbits=mask_bits if mask_src is not None else None, wt_rows=wrows)
    if bn_y is None:
        return dx
    return (dx, partial) if bn_y2 is None else (dx, partial, partial2)


def _conv_dgrad_bn(dy, w, x_shape, stride, padding, y_prev, fcoef_prev, dy_y=None, dy_coef=None):
    """Backward-data whose output is the gradient of ``relu(bn(y_prev))``: the kernel stores the
    ReLU-masked gradient g and the per-tile partials of sum g / sum g*y_prev, i.e. the reduce
    pass of that BatchNorm's backward (K8a) rides in the epilogue.  Returns (g, partial).
    ``dy_y`` + ``dy_coef`` (only where :func:`_rows_bn_bwd_ok`): ``dy`` is the gradient w.r.t. the output of the
    BatchNorm behind this conv and the kernel forms ca*dy + cb*dy_y + cc itself."""
    B, C, H, W = x_shape
    _, _, R, S = w.shape
    wt, wrows = _dgrad_w(w)
    g = _empty_nhwc(B, C, H, W, dy.device)
    if wrows == 0 and w.shape[0] == C and _rows_ok(dy, w, stride, padding) and \
            y_prev.stride() == g.stride():
        partial = torch.empty((B, 2, C), dtype=torch.float32, device=dy.device)   # one row per image
        K.conv_rows3x3(dy, wt, g, stats=partial, ep_y=y_prev, ep_fcoef=fcoef_prev, dy_y=dy_y, dy_coef=dy_coef)
        return g, partial
    if dy_y is not None:
        raise K.KernelError("conv_dgrad_bn: only the row-streaming kernel forms its input gradient")
    T = K.conv_tiles_m_bwd(B * H * W, H, W, stride, 1)
    partial = torch.empty((T, 2, C), dtype=torch.float32, device=dy.device)
    K.conv_igemm(dy, _geom(dy), wt, g, C, R, S, 1, R - 1 - padding, S - 1 - padding, stride, H, W,
                 stats=partial, ep_y=y_prev, ep_fcoef=fcoef_prev, wt_rows=wrows)
    return g, partial


def _conv_dgrad_1x1s2_compact(dy, w):
    """Gradient of a 1x1 / stride 2 conv at the even input pixels only: [B, Cin, Ho, Wo]."""
    B, Kc, Ho, Wo = dy.shape
    C = w.shape[1]
    wt, wrows = _dgrad_w(w)
    out = _empty_nhwc(B, C, Ho, Wo, dy.device)
    K.conv_igemm(dy, _geom(dy), wt, out, C, 1, 1, 1, 0, 0, 1, Ho, Wo, wt_rows=wrows)
    return out


_PROBE_NO_PREWAIT = _os_rows.environ.get("DDL_PROBE_NO_PREWAIT", "0") == "1"


def _side_prewait(device):
    """Order the weight-gradient stream behind what is queued on the current stream NOW and return
    True, so that the weight-gradient kernels themselves can be issued later (after the next kernel
    of the dependent chain) without waiting for that kernel too.  False: no lean side stream."""
    if not (_SIDE["on"] and _SIDE_LEAN and _DIRECT["on"] and _grad_listeners is not None):
        return False
    side = _side_stream(device)
    if _PROBE_NO_PREWAIT:   # timing-only probe (results are wrong): what the event records on the main stream cost
        return True
    K.stream_wait_stream(side.cuda_stream, K._raw_stream(), _SIDE["event"])
    return True


def _conv_wgrad(dy, x, w, stride, padding, prewaited=False, x_fcoef=None, dy_y=None, dy_coef=None, keep=()):
    """dw for ``w``: written into ``w.grad`` when that is a live arena view (returns None and
    notifies the listeners), otherwise returned as a new tensor.

    In direct mode the kernel pair (split-K partials + ordered reduce) runs on a second HIP
    stream: weight gradients are off the critical path of backward (nothing downstream in the
    backward pass reads them), so they fill the bubbles the dependent chain dgrad -> BN finalize
    -> BN apply leaves on the main stream.  ``keep``: tensors behind raw addresses among the arguments
    (``x_fcoef`` points into a SAVED coefficient block) - the side stream reads them after the autograd node
    has returned and released its saved tensors, so they are held until the join like ``dy`` and ``x``."""
    Kc, _, R, S = w.shape
    Ho, Wo = dy.shape[2], dy.shape[3]
    if _grad_listeners is not None and _direct_grad_ok(w) and _DIRECT["on"]:
        side = _side_stream(dy.device) if (_SIDE["on"] and Ho * Wo < _SIDE_MAX_HW) else None
        if side is None:
            _wgrad_into(dy, x, _geom(x), w.grad, Kc, R, S, stride, padding, padding, Ho, Wo,
                        accumulate=True, x_fcoef=x_fcoef, dy_y=dy_y, dy_coef=dy_coef)
        else:
            # raw event + raw stream handle: no torch stream context, no torch Event objects and a
            # persistent partial-sum workspace (side-stream kernels run in order, one buffer
            # serves them all) - ~25 us of host time per weight gradient less
            if _SIDE_LEAN:
                raw = side.cuda_stream
                if not prewaited:
                    K.stream_wait_stream(raw, K._raw_stream(), _SIDE["event"])
                _wgrad_into(dy, x, _geom(x), w.grad, Kc, R, S, stride, padding, padding, Ho, Wo,
                            accumulate=True, stream=raw, workspace=_side_workspace, x_fcoef=x_fcoef,
                            dy_y=dy_y, dy_coef=dy_coef)
            else:  # torch's own stream switch and events (DDL_SIDE_LEAN=0)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    _wgrad_into(dy, x, _geom(x), w.grad, Kc, R, S, stride, padding, padding, Ho, Wo,
                                accumulate=True, x_fcoef=x_fcoef, dy_y=dy_y, dy_coef=dy_coef)
            # The side stream reads dy and x after this function returns.  They are kept alive
            # until the join at the end of backward instead of `record_stream`: blocks with a
            # recorded stream are withheld from reuse until that stream has passed an event, and
            # with the host several steps ahead of the GPU the allocator then keeps calling
            # hipMalloc for new segments (reserved memory 27 -> 55-70 GB, and in 8 % of the bench
            # processes a 1-1.8 s burst of device allocations inside the timed loop -
            # profiles/r1_notes.md).  Released after the join is queued, they go back to this
            # stream's pool and any later kernel that receives them is ordered after the side
            # stream's reads.
            _side_mark_used()
            if _SIDE["join_queued"]:
                _SIDE["keep"].append((dy, x, dy_y, dy_coef) + tuple(keep))
            else:  # raw call outside a backward pass: nobody is known to join, let the allocator track it
                dy.record_stream(side)
                x.record_stream(side)
                for t in (dy_y, dy_coef) + tuple(keep):
                    if torch.is_tensor(t):
                        t.record_stream(side)
        _notify(w)
        return None
    return _wgrad_into(dy, x, _geom(x), torch.empty_like(w), Kc, R, S, stride, padding, padding,
                       Ho, Wo, x_fcoef=x_fcoef, dy_y=dy_y, dy_coef=dy_coef)


# ---- second stream for weight gradients --------------------------------------------------------
import os as _os
_SIDE = {"on": _os.environ.get("DDL_WGRAD_SIDE_STREAM", "1") != "0",   # 0: weight gradients on the main stream
         "stream": None, "pending": False, "event": None, "ws": None, "keep": [], "join_queued": False}
_SIDE_LEAN = _os.environ.get("DDL_SIDE_LEAN", "1") != "0"
# weight gradients of layers with at least this many output pixels per image run on the MAIN stream
# (probe: do the bandwidth-bound early layers gain anything from a second stream?)
_SIDE_MAX_HW = int(_os.environ.get("DDL_SIDE_MAX_HW", str(1 << 30)))


def _side_workspace(nbytes):
    """Address of the side stream's split-K workspace (grown on demand; allocated under the side
    stream so that the allocator recycles a replaced buffer in that stream's order)."""
    ws = _SIDE["ws"]
    if ws is None or ws.numel() < nbytes:
        side = _SIDE["stream"]
        with torch.cuda.stream(side):
            ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=side.device)
        _SIDE["ws"] = ws
    return ws.data_ptr()


def set_wgrad_side_stream(on: bool):
    _SIDE["on"] = bool(on)


def _side_stream(device):
    """The weight-gradient stream: created through HIP with the LOWEST priority the device offers.
    A priority of its own also gives it a hardware queue of its own (streams of equal priority
    share the few HIP hardware queues round-robin, and two streams on one queue serialise — a
    pool stream from torch landed on the default stream's queue, profiles/r1_notes.md), and the
    dependent main-stream chain wins when both have workgroups to place."""
    st = _SIDE["stream"]
    if st is None or st.device != device:
        import os

        prio = os.environ.get("DDL_WGRAD_STREAM_PRIORITY")
        if prio is None:
            # Several ranks on ONE device over the HOST backend (gloo rehearsal rig; never the production layout): a
            # low-priority queue of one process can be starved for ever by a normal-priority queue of ANOTHER process that
            # sits on a wait which only this process can end (priority inversion across processes - an intermittent hang of
            # the two-rank gloo rehearsal with the GPU 100 % busy, profiles/r3_notes.md section 28): normal priority there.
            # (The two-rank xGMI rig needs the opposite: its resident collective kernels run at low priority, and a
            # normal-priority weight-gradient stream next to them stalls the exchange - it keeps `low`.)
            try:
                shared = int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > max(torch.cuda.device_count(), 1) and \
                    os.environ.get("HVD_BACKEND", "") == "gloo"
            except ValueError:
                shared = False
            prio = "0" if shared else "low"
        if prio == "torch":
            st = torch.cuda.Stream(device=device)
        else:
            with torch.cuda.device(device):
                least, greatest = K.stream_priority_range()
                want = {"low": least, "high": greatest}.get(prio)
                if want is None:
                    want = int(prio)
                st = torch.cuda.ExternalStream(K.create_stream(want), device=device)
        with torch.cuda.device(device):
            _SIDE["event"] = K.create_event()
            _SIDE["event_wt"] = K.create_event()
            _SIDE["event_join"] = K.create_event()
        _SIDE["stream"] = st
        _SIDE["ws"] = None
    return st


def _join_side_stream():
    """Order the current stream after everything queued on the weight-gradient stream."""
    st = _SIDE["stream"]
    if st is not None and _SIDE["pending"]:
        cur = torch.cuda.current_stream(st.device)
        K.stream_wait_stream(cur.cuda_stream, st.cuda_stream, _SIDE["event_join"])
        main = _SIDE.get("main")
        if main is not None and main != cur:  # the stream backward ran on
            K.stream_wait_stream(main.cuda_stream, st.cuda_stream, _SIDE["event_join"])
    _SIDE["pending"] = False
    _SIDE["join_queued"] = False
    if _SIDE["keep"]:
        _SIDE["keep"] = []   # operands of the queued weight gradients: free to be reused from here on


def _side_mark_used():
    # the join is queued once per backward pass: it runs when autograd finishes the pass, so
    # whoever reads param.grad afterwards (optimizer, clipping, tests) is ordered correctly
    if not _SIDE["pending"]:
        _SIDE["pending"] = True
        _SIDE["main"] = torch.cuda.current_stream()
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_side_stream)
            _SIDE["join_queued"] = True
        except RuntimeError:  # not inside a backward pass (raw calls in tests / benchmarks)
            _SIDE["join_queued"] = False


class grad_exchange_context:
    """Context for launching a gradient collective from a readiness notification: the launch
    happens on the weight-gradient stream after it has been ordered behind the main stream, so
    the collective sees the gradients written on either of them."""

    def __enter__(self):
        st = _SIDE["stream"]
        self.cm = None
        if st is not None and _SIDE["pending"]:
            st.wait_stream(torch.cuda.current_stream(st.device))
            self.cm = torch.cuda.stream(st)
            self.cm.__enter__()
        return self

    def __exit__(self, *exc):
        if self.cm is not None:
            self.cm.__exit__(*exc)
        return False


_DIRECT = {"on": False}


def _bn_fwd(y, gamma, beta, residual, stats, rmean, rvar, training, momentum, eps, relu):
    B, C, H, W = y.shape
    M = B * H * W
    dev = y.device
    coeffs = torch.empty((4, C), dtype=torch.float32, device=dev)  # mean, invstd, scale, shift
    if training:
        if stats is None or stats.numel() == 0:
            rows = 512
            T = (M + rows - 1) // rows
            stats = torch.empty((T, 2, C), dtype=torch.float32, device=dev)
            K.bn_stats_partial(y, stats, M, C, rows)
        else:
            T = stats.shape[0]
            rows = 128 if (T - 1) * 128 < M <= T * 128 else M // T   # (row-streaming conv: one row per image)
        K.bn_fwd_finalize(stats, T, C, M, rows, gamma, beta, rmean, rvar, momentum, eps,
                          coeffs[0], coeffs[1], coeffs[2], coeffs[3])
    else:
        K.bn_eval_coeffs(gamma, beta, rmean, rvar, eps, coeffs[2], coeffs[3])
    z = torch.empty_like(y)
    K.bn_apply_fwd(y, coeffs[2], coeffs[3], residual, z, M, C, relu)
    return z, coeffs


def _bn_bwd(dz, y, z, gamma, coeffs, relu, want_g, beta=None, has_res=True, partial=None,
            premasked=False):
    """(dy, dgamma, dbeta, g) with g = dz masked by the ReLU (None unless want_g and relu).

    With ``beta`` given and direct gradients enabled, dgamma/dbeta are accumulated by the
    finalize kernel straight into ``gamma.grad`` / ``beta.grad`` (returned as None).
    ``partial``: the reduce pass already ran in the producing kernel's epilogue (``dz`` is then
    the masked gradient).  ``premasked``: ``dz`` already carries the ReLU mask."""
    B, C, H, W = y.shape
    M = B * H * W
    dev = y.device
    # ReLU mask: from the stored activation when a residual was added before the ReLU, else
    # recomputed from y with the forward scale/shift (z is then never read)
    mask = 0 if (not relu or premasked or partial is not None) else (1 if has_res else 2)
    fco = coeffs[2:4] if mask == 2 else None
    if partial is None:
        G = K.bn_bwd_blocks(M)
        partial = torch.empty((G, 2, C), dtype=torch.float32, device=dev)
        K.bn_bwd_reduce(dz, y, z if mask == 1 else None, fco, partial, M, C, mask, G)
    else:
        G = partial.shape[0]
    coef = torch.empty((3, C), dtype=torch.float32, device=dev)
    direct = beta is not None and _DIRECT["on"] and _direct_grad_ok(gamma) and \
        _direct_grad_ok(beta) and gamma.grad.dtype == torch.float32 and \
        beta.grad.dtype == torch.float32
    if direct:
        K.bn_bwd_finalize(partial, G, C, M, gamma, coeffs[0], coeffs[1], gamma.grad, beta.grad,
                          coef, accumulate=True)
        dg = db = None
    else:
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        K.bn_bwd_finalize(partial, G, C, M, gamma, coeffs[0], coeffs[1], dgb[0], dgb[1], coef)
        dg, db = dgb[0], dgb[1]
    dy = torch.empty_like(y)
    g = torch.empty_like(y) if (want_g and mask != 0) else None
    K.bn_bwd_apply(dz, y, z if mask == 1 else None, fco, coef, dy, g, M, C, mask)
    if want_g and mask == 0:
        g = dz
    if direct:
        _notify(gamma)
        _notify(beta)
    return dy, dg, db, g


# ---- lean conv-BN unit used by the fused blocks ---------------------------------------------------
# Same kernels as _conv_fwd/_bn_fwd/_bn_bwd, but per-layer scratch (statistics partials, fold
# workspace, backward coefficients) lives in ONE allocation addressed by raw pointers, and the
# saved [4, C] coefficient block is addressed the same way: each avoided tensor object (an
# allocation or a view) is 3-5 us of host time, ~30 of them per layer and step, and the host has
# to keep ahead of ~50 us kernels in the late layers (profiles/r1_notes.md).
def _unit_fwd(x, w, stride, padding, gamma, beta, residual, rmean, rvar, momentum, eps, relu,
              apply=True, res_coeffs=None, bits=None, keep=None, xf=None):
    """Training-mode conv -> BatchNorm(+residual)(+ReLU).  Returns (y, z, coeffs[4, C]).
    ``apply=False`` stops after the statistics (z is None): the caller folds this BatchNorm into
    a later kernel.  ``res_coeffs``: coefficient block of such a deferred BatchNorm whose raw conv
    output is passed as ``residual``.  ``bits``: uint8 buffer that receives ``z > 0`` bitwise."""
    B, C, H, W = x.shape
    Kc, Cw, R, S = w.shape
    if Cw != C:
        raise K.KernelError("conv: input has %d channels, weight expects %d" % (C, Cw))
    Ho = (H + 2 * padding - R) // stride + 1
    Wo = (W + 2 * padding - S) // stride + 1
    M = B * Ho * Wo
    rows = _rows_ok(x, w, stride, padding)
    T = B if rows else K.conv_tiles_m(M)      # (row-streaming kernel: one statistics row per image)
    P = K._fold_slices(T) if T > K.bn_direct_max() else 0
    dev = x.device
    y = torch.empty((B, Kc, Ho, Wo), dtype=_BF16, device=dev, memory_format=_CL)
    coeffs = torch.empty((4, Kc), dtype=torch.float32, device=dev)  # mean, invstd, scale, shift
    tmp = torch.empty(2 * T * Kc + 4 * P * Kc, dtype=torch.float32, device=dev)
    st = tmp.data_ptr()
    wk = st + 8 * T * Kc if P else None      # fp64 fold workspace (8-byte aligned: T*Kc*8 bytes in)
    if xf is not None:
        # x is the RAW output of the previous unit's conv (its BatchNorm was left unapplied): the row-streaming
        # kernel applies scale / shift / ReLU to the rows it has staged and writes them out on the side
        in_coeffs, z_in = xf
        K.conv_rows3x3(x, w, y, stats=st, in_fcoef=in_coeffs.data_ptr() + 8 * C, z_out=z_in)
    elif rows:
        K.conv_rows3x3(x, w, y, stats=st)
    else:
        K.conv_igemm(x, _geom(x), w, y, Kc, R, S, stride, padding, padding, 1, Ho, Wo, stats=st)
    cb = coeffs.data_ptr()
    K.bn_fwd_finalize(st, T, Kc, M, Ho * Wo if rows else 128, gamma, beta, rmean, rvar, momentum, eps,
                      cb, cb + 4 * Kc, cb + 8 * Kc, cb + 12 * Kc, work=wk)
    if keep is not None:   # launched on another stream: the caller holds the scratch until the streams are ordered
        keep.append(tmp)
    if not apply:
        return y, None, coeffs
    z = torch.empty_like(y)
    if res_coeffs is not None:
        rb = res_coeffs.data_ptr()
        K.bn_apply_fwd(y, cb + 8 * Kc, cb + 12 * Kc, residual, z, M, Kc, relu,
                       rb + 8 * Kc, rb + 12 * Kc, bits=bits)
    else:
        K.bn_apply_fwd(y, cb + 8 * Kc, cb + 12 * Kc, residual, z, M, Kc, relu, bits=bits)
    return y, z, coeffs


def _unit_bwd(dz, y, z, gamma, beta, coeffs, mask, want_g, partial=None):
    """BatchNorm(+ReLU) backward of one unit.  ``mask``: 0 none (``dz`` is final), 1 from ``z``,
    2 recomputed from ``y``.  ``partial`` = reduce partials already produced by a conv epilogue.
    Returns (dy, dgamma, dbeta, g) like _bn_bwd."""
    B, C, H, W = y.shape
    M = B * H * W
    dev = y.device
    cb = coeffs.data_ptr()
    fco = cb + 8 * C if mask == 2 else None
    if partial is None:
        G = K.bn_bwd_blocks(M)
        own = 2 * G * C
    else:
        G = partial.shape[0]
        own = 0
    P = K._fold_slices(G) if G > K.bn_direct_max() else 0
    tmp = torch.empty(3 * C + own + 2 * P * C, dtype=torch.float32, device=dev)
    coef = tmp.data_ptr()
    if partial is None:
        part = coef + 12 * C
        K.bn_bwd_reduce(dz, y, z if mask == 1 else None, fco, part, M, C, mask, G)
    else:
        part = partial.data_ptr()
    wk = coef + 12 * C + 4 * own if P else None
    direct = _DIRECT["on"] and _direct_grad_ok(gamma) and _direct_grad_ok(beta) and \
        gamma.grad.dtype == torch.float32 and beta.grad.dtype == torch.float32
    if direct:
        K.bn_bwd_finalize(part, G, C, M, gamma, cb, cb + 4 * C, gamma.grad, beta.grad, coef,
                          accumulate=True, work=wk)
        dg = db = None
    else:
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        K.bn_bwd_finalize(part, G, C, M, gamma, cb, cb + 4 * C, dgb[0], dgb[1], coef, work=wk)
        dg, db = dgb[0], dgb[1]
    dy = torch.empty_like(y)
    g = torch.empty_like(y) if (want_g and mask != 0) else None
    if _PROBE_SKIP_WIDE_APPLY and want_g and mask == 0:
        pass  # timing-only probe (results are wrong): price of the wide unit's backward apply pass
    else:
        K.bn_bwd_apply(dz, y, z if mask == 1 else None, fco, coef, dy, g, M, C, mask)
    if want_g and mask == 0:
        g = dz
    if direct:
        _notify(gamma)
        _notify(beta)
    # tmp is only referenced by raw address from here on: the launches above are already queued
    # on this stream and the allocator hands the block out again in stream order
    return dy, dg, db, g


# ---- BatchNorm backward of a bottleneck's last unit by linearity -----------------------------------
# dy3 = ca*g + cb*y3 + cc per channel and y3 = z2 W3^T (1x1 conv), hence
#   dz2 = g (diag(ca) W3) + z2 (W3^T diag(cb) W3) + 1 (cc^T W3)                    (input gradient)
#   dW3 = diag(ca) (g^T z2) + diag(cb) W3 (z2^T z2) + cc (x) colsum(z2)             (weight gradient;
#         evaluated with the CENTRED Gram matrix z2^T z2 - colsum (x) colsum / M, see splitk_reduce_lin_kernel)
# so dy3 - three passes over the block's WIDEST tensor (read g, read y3, write dy3, then read twice
# more by the two GEMMs) - never exists.  The input gradient becomes ONE two-segment GEMM over g and
# z2 (K grows by a quarter); the weight gradient is the raw product g^T z2 plus a Gram matrix of the
# narrow activation, fixed up in the split-K reduction.  Costed in profiles/r1_notes.md (the apply
# pass alone: 1.57 ms of the 19.2 ms step); numerics in benchmarks/bn_linearity_check.py and the
# block / trajectory tests.  DDL_BN3_LINEAR=0 restores the materialised dy3.
_BN3_LINEAR = {"on": _os.environ.get("DDL_BN3_LINEAR", "1") != "0",
               "max_c2": int(_os.environ.get("DDL_BN3_LINEAR_MAX_C2", "256"))}   # layer 4 (512 -> 2048 at 7x7) measured faster with dy3 materialised
# the same for the projection shortcut's BatchNorm, where the conv's input is narrow enough for the
# Cin x Cin Gram / Q matrices and the longer reduction to cost less than the pass they replace
_DS_LINEAR = {"on": _os.environ.get("DDL_DS_LINEAR", "1") != "0",
              "max_cin": int(_os.environ.get("DDL_DS_LINEAR_MAX_CIN", "64"))}


_LIN_PREP_Q = int(_os.environ.get("DDL_LIN_PREP_Q", "1"))
# ... for these narrow widths.  Stage 3 (256) keeps the two launches: in the step the slow 4-workgroup GEMM is the
# window in which the side stream's 256 x 256 weight-gradient workgroups (a whole CU each) get placed; without it
# the step is 0.13 ms SLOWER although the chain is 0.1 ms shorter (profiles/r3_notes.md section 13)
_LIN_PREP_Q_C2 = [int(v) for v in _os.environ.get("DDL_LIN_PREP_Q_C2", "64:128").split(":") if v]
# the side stream starts a linear unit's raw weight-gradient product before the main stream folds / finalizes its coefficients
_LIN_EARLY_SIDE = int(_os.environ.get("DDL_LIN_EARLY_SIDE", "0"))   # 2: its launches are issued first as well   # 0: the two-launch form (A/B switch)


def _unit_coef(dz, y, gamma, beta, coeffs, partial):
    """Finalize pass only of a BatchNorm backward whose reduce partials already exist: returns the
    scratch tensor whose first 3*C floats are (ca, cb, cc); dgamma/dbeta go where _unit_bwd puts
    them (returned as (tmp, dg, db))."""
    B, C, H, W = y.shape
    M = B * H * W
    dev = y.device
    cb = coeffs.data_ptr()
    G = partial.shape[0]
    P = K._fold_slices(G) if G > K.bn_direct_max() else 0
    tmp = torch.empty(3 * C + 2 * P * C, dtype=torch.float32, device=dev)
    coef = tmp.data_ptr()
    wk = coef + 12 * C if P else None
    direct = _DIRECT["on"] and _direct_grad_ok(gamma) and _direct_grad_ok(beta) and \
        gamma.grad.dtype == torch.float32 and beta.grad.dtype == torch.float32
    if direct:
        K.bn_bwd_finalize(partial.data_ptr(), G, C, M, gamma, cb, cb + 4 * C, gamma.grad, beta.grad, coef,
                          accumulate=True, work=wk)
        dg = db = None
        _notify(gamma)
        _notify(beta)
    else:
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        K.bn_bwd_finalize(partial.data_ptr(), G, C, M, gamma, cb, cb + 4 * C, dgb[0], dgb[1], coef, work=wk)
        dg, db = dgb[0], dgb[1]
    return tmp, dg, db


def _bn3_linear_ok(w, stride, padding, g, z2):
    Kc, C2, R, S = w.shape
    return _BN3_LINEAR["on"] and not _WT_KRSC and R == 1 and S == 1 and stride == 1 and padding == 0 and \
        C2 % 64 == 0 and Kc % 32 == 0 and Kc <= 2048 and g.shape[1] == Kc and z2.shape[1] == C2 and \
        g.shape[2:] == z2.shape[2:] and C2 <= _BN3_LINEAR["max_c2"]


def _lin_main_static_ok(w, stride, padding):
    Kc, C2, R, S = w.shape
    return _BN3_LINEAR["on"] and not _WT_KRSC and R == 1 and S == 1 and stride == 1 and padding == 0 and \
        C2 % 64 == 0 and Kc % 32 == 0 and Kc <= 2048 and C2 <= _BN3_LINEAR["max_c2"]


def _lin_ds_static_ok(wd, ds_stride, x):
    even = x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
    return _DS_LINEAR["on"] and _BN3_LINEAR["on"] and not _WT_KRSC and (ds_stride == 1 or (ds_stride == 2 and even)) and \
        wd.shape[2] == 1 and wd.shape[1] % 64 == 0 and wd.shape[0] % 32 == 0 and wd.shape[0] <= 2048 and \
        wd.shape[1] <= _DS_LINEAR["max_cin"]


# the input gradient and the raw weight-gradient product g^T z2 of a linear unit in ONE pass over g
# (csrc/kernels/lin_fused.hip; DDL_LIN_FUSED=0: the two-segment GEMM + a separate weight-gradient launch)
# "gram" (DDL_LIN_FUSED_GRAM=1): the fused kernel also accumulates z2^T z2 and the column sums of z2 (stage 1), which
# removes the Gram launch from the weight-gradient stream: 0.4 GB of traffic and four launches less per step, step time
# EQUAL (16.17 / 16.16 ms: that launch ran in the shadow of the chain) - off by default, the longer-serving path stays.
_LIN_FUSED = {"on": _os.environ.get("DDL_LIN_FUSED", "1") != "0",
              "gram": _os.environ.get("DDL_LIN_FUSED_GRAM", "0") == "1"}


def _bn3_linear_dgrad(g, z2, w, coef_t, y_prev, fcoef_prev, fuse=None):
    """Input gradient of the 1x1 conv behind the BatchNorm whose backward coefficients are in
    ``coef_t``: g [B, C, H, W] is that BatchNorm's (ReLU-masked) output gradient, z2 the conv's
    input.  Returns (g2, partial) like _conv_dgrad_bn: the ReLU-masked gradient of the previous
    unit's BatchNorm output and its backward partial sums.  ``fuse`` (a dict, or None): where the fused kernel
    serves the shape it also computes the slabs of g^T z2 and leaves them in ``fuse["P"]`` ([splits, C*C2] fp32,
    ``fuse["splits"]``) for :func:`_bn3_linear_wgrad_start`."""
    B, C, H, W = g.shape
    C2 = z2.shape[1]
    dev = g.device
    wt3 = _WT.get(w)                      # [C2][C]: wt3[k][c] = w[c][k]
    bias = torch.empty(C2, dtype=torch.float32, device=dev)
    if _LIN_PREP_Q > 0 and (not _LIN_PREP_Q_C2 or C2 in _LIN_PREP_Q_C2):
        # b1, bias and Q[k][j] = sum_c cb[c] w[c][k] w[c][j] in one launch (the dependent chain waits for it)
        if _LIN_PREP_Q == 2:
            scratch = torch.empty(2 * C2 * C + C2 * C2, dtype=_BF16, device=dev)
            b1, q = scratch[:C2 * C], scratch[2 * C2 * C:]
        else:
            scratch = torch.empty(C2 * C + C2 * C2, dtype=_BF16, device=dev)
            b1, q = scratch[:C2 * C], scratch[C2 * C:]
        K.bn_lin_prep_q(wt3, C, coef_t.data_ptr(), C, C2, b1, q, bias)
    else:
        scratch = torch.empty(2 * C2 * C + C2 * C2, dtype=_BF16, device=dev)
        b1, a2, q = scratch[:C2 * C], scratch[C2 * C: 2 * C2 * C], scratch[2 * C2 * C:]
        K.bn_lin_prep(wt3, C, coef_t.data_ptr(), C, C2, b1, a2, bias)
        # Q: a2 (rows k) times wt3^T, a C2 x C2 GEMM over C
        K.conv_igemm(a2, (C2, 1, 1, C, C, C, C, a2.numel()), wt3, q, C2, 1, 1, 1, 0, 0, 1, 1, 1)
        if _LIN_PREP_Q < 0:   # timing probe: the chain's price of this GEMM (issued twice)
            K.conv_igemm(a2, (C2, 1, 1, C, C, C, C, a2.numel()), wt3, q, C2, 1, 1, 1, 0, 0, 1, 1, 1)
    g2 = _empty_nhwc(B, C2, H, W, dev)
    M = B * H * W
    nwg = K.lin_fused_wgs(M, C, C2) if (fuse is not None and _LIN_FUSED["on"]) else 0
    if nwg > 0 and g.stride() == (H * W * C, 1, W * C, C) and z2.stride() == (H * W * C2, 1, W * C2, C2) and \
            tuple(z2.shape[2:]) == (H, W) and (y_prev is None or y_prev.stride() == z2.stride()):
        partial = None if y_prev is None else torch.empty((nwg, 2, C2), dtype=torch.float32, device=dev)
        P = torch.empty((nwg, C * C2), dtype=torch.float32, device=dev)
        G = ZS = None
        if fuse.get("want_gram") and _LIN_FUSED["gram"] and K.lin_fused_gram_ok(C, C2):
            # the forward-data part of the weight gradient (z2^T z2, column sums of z2) from the same z2 tiles
            G = torch.empty((nwg, C2 * C2), dtype=torch.float32, device=dev)
            ZS = torch.empty((nwg, 2, C2), dtype=torch.float32, device=dev)
        K.lin_fused(g, z2, b1, q, bias, y_prev, fcoef_prev, g2, partial, P, M, C, C2, nwg, gram=G, colsum=ZS)
        fuse["P"], fuse["splits"], fuse["G"], fuse["ZS"] = P, nwg, G, ZS
        return g2, partial
    if y_prev is None:   # plain store (projection shortcut: the result is the addend of the block-input gradient)
        K.conv_igemm(g, _geom(g), b1, g2, C2, 1, 1, 1, 0, 0, 1, H, W, x2=z2, w2=q, c2=C2, bias=bias)
        return g2, None
    T = K.conv_tiles_m_bwd(M, H, W, 1, 1)
    partial = torch.empty((T, 2, C2), dtype=torch.float32, device=dev)
    K.conv_igemm(g, _geom(g), b1, g2, C2, 1, 1, 1, 0, 0, 1, H, W, stats=partial, ep_y=y_prev,
                 ep_fcoef=fcoef_prev, x2=z2, w2=q, c2=C2, bias=bias)
    return g2, partial


def _lin_gram_layout(M, C, C2):
    rows = 512
    Tz = (M + rows - 1) // rows
    s2 = K.conv_wgrad_splits(M, C2, C2, 1, 1)
    return rows, Tz, s2


def _bn3_linear_gram(z2, w, raw, base, t, colz, slabs=None):
    """The part of the linear weight gradient that only depends on FORWARD data: column sums of z2,
    the centred Gram matrix z2^T z2 - colz (x) colz / M and T = W (that matrix), written to the raw
    addresses ``colz`` ([2][C2] fp32) and ``t`` ([C][C2] bf16); ``base`` = scratch on stream ``raw``."""
    C, C2 = w.shape[0], w.shape[1]
    B, _, H, W = z2.shape
    M = B * H * W
    gz = _geom(z2)
    rows, Tz, s2 = _lin_gram_layout(M, C, C2)
    gslab = C2 * C2
    gp = base
    gram = gp + (4 * s2 * gslab + 255) // 256 * 256
    zp = gram + (2 * gslab + 255) // 256 * 256
    if slabs is not None:
        # (G, ZS, n): the Gram slabs and column sums came out of the fused input-gradient kernel
        G, ZS, n = slabs
        K.bn_colsum_fold(ZS, n, C2, colz, stream=raw)
        K.splitk_reduce_centered(G, gram, C2, n, colz, M, stream=raw)
        K.conv_igemm(w, (C, 1, 1, C2, C2, C2, C2, w.numel()), gram, _RawOut(t, C * C2), C2, 1, 1, 1, 0, 0, 1, 1, 1,
                     stream=raw, w_numel=gslab)
        return
    if K.conv_wgrad_colsum_ok(M, C2, C2):
        # the Gram launch sums the columns of z2 on the side (a row of ones through the matrix pipe): no extra pass over z2
        K.conv_wgrad(z2, z2, gz, gp, C2, 1, 1, 1, 0, 0, H, W, s2, stream=raw, colsum=zp)
        K.bn_colsum_fold(zp, s2, C2, colz, stream=raw)
    else:
        K.conv_wgrad(z2, z2, gz, gp, C2, 1, 1, 1, 0, 0, H, W, s2, stream=raw)
        K.bn_stats_partial(z2, zp, M, C2, rows, stream=raw)
        K.bn_colsum_fold(zp, Tz, C2, colz, stream=raw)
    K.splitk_reduce_centered(gp, gram, C2, s2, colz, M, stream=raw)   # z2^T z2 - colz (x) colz / M
    # T = W3 (centred Gram): rows c of the filter times the (symmetric) matrix
    K.conv_igemm(w, (C, 1, 1, C2, C2, C2, C2, w.numel()), gram, _RawOut(t, C * C2), C2, 1, 1, 1, 0, 0, 1, 1, 1,
                 stream=raw, w_numel=gslab)


def _lin_gram_scratch_bytes(M, C, C2):
    rows, Tz, s2 = _lin_gram_layout(M, C, C2)
    gslab = C2 * C2
    return (4 * s2 * gslab + 255) // 256 * 256 + (2 * gslab + 255) // 256 * 256 + 4 * max(Tz, s2) * 2 * C2 + 256


# The forward pass leaves the weight-gradient stream idle: the Gram part of a block that will take
# the linear path in backward runs there, right after the block's narrow activation exists (it is
# still in the Infinity Cache then), instead of in the crowded backward window.  Measured SLOWER
# (18.51 / 18.52 ms against 18.44 / 18.45 ms: the forward pass is main-stream time too and slows down
# by what backward gains), so it is opt-in: DDL_LIN_GRAM_FWD=1.
_LIN_GRAM_FWD = {"on": _os.environ.get("DDL_LIN_GRAM_FWD", "0") == "1"}


def _bn3_linear_gram_forward(z2, w):
    """Forward-time launch of :func:`_bn3_linear_gram` on the weight-gradient stream.  Returns the
    tensor holding T ([C][C2] bf16) followed by the column sums ([2][C2] fp32), or None when the
    side stream is not in use."""
    if not (_LIN_GRAM_FWD["on"] and _SIDE["on"] and _SIDE_LEAN and _DIRECT["on"] and _grad_listeners is not None):
        return None
    if torch.cuda.is_current_stream_capturing():
        return None
    C, C2 = w.shape[0], w.shape[1]
    B, _, H, W = z2.shape
    M = B * H * W
    tbytes = (2 * C * C2 + 255) // 256 * 256
    pre = torch.empty(tbytes + 4 * 2 * C2, dtype=torch.uint8, device=z2.device)
    side = _side_stream(z2.device)
    raw = side.cuda_stream
    K.stream_wait_stream(raw, K._raw_stream(), _SIDE["event"])
    base = _side_workspace(_lin_gram_scratch_bytes(M, C, C2))
    _bn3_linear_gram(z2, w, raw, base, pre.data_ptr(), pre.data_ptr() + tbytes)
    return pre


def _bn3_linear_wgrad_start(g, z2, w, prewaited=False, pre=None, raw_product=None):
    """First half of :func:`_bn3_linear_wgrad`: everything that needs no BatchNorm coefficients (the raw
    product g^T z2 - unless ``raw_product`` = {"P", "splits"} brings it from the fused input-gradient kernel; the
    second half must then be ordered behind the main stream once more - and, unless ``pre`` has it, the Gram part).
    Returns the state for the second half."""
    C, C2 = w.shape[0], w.shape[1]
    direct = _grad_listeners is not None and _direct_grad_ok(w) and _DIRECT["on"]
    out = w.grad if direct else torch.zeros_like(w)
    side = _side_stream(g.device) if (_SIDE["on"] and direct and g.shape[2] * g.shape[3] < _SIDE_MAX_HW) else None
    B, _, H, W = g.shape
    M = B * H * W
    rawP = raw_product["P"] if raw_product else None
    rawG = raw_product.get("G") if raw_product else None
    splits = raw_product["splits"] if rawP is not None else K.conv_wgrad_splits(M, C, C2, 1, 1)      # raw product g^T z2
    slab = C * C2
    pbytes = 0 if rawP is not None else (4 * splits * slab + 255) // 256 * 256
    tbytes = (2 * slab + 255) // 256 * 256
    total = pbytes + (0 if pre is not None else tbytes + 4 * 2 * C2 + 256 + _lin_gram_scratch_bytes(M, C, C2))
    ws_t = None
    if side is None:
        ws_t = torch.empty(total, dtype=torch.uint8, device=g.device)
        base, raw = ws_t.data_ptr(), None
    else:
        raw = side.cuda_stream
        if not prewaited or rawG is not None:      # (the Gram slabs are written by the main stream's fused kernel)
            K.stream_wait_stream(raw, K._raw_stream(), _SIDE["event"])
        base = _side_workspace(total)
    if rawP is not None:
        partial = rawP.data_ptr()
    else:
        partial = base
        K.conv_wgrad(g, z2, _geom(z2), partial, C, 1, 1, 1, 0, 0, H, W, splits, stream=raw)
    if pre is not None:
        t = pre.data_ptr()
        colz = t + tbytes
    else:
        t = base + pbytes
        colz = t + tbytes
        _bn3_linear_gram(z2, w, raw, colz + 8 * C2, t, colz,   # (8*C2 bytes of column sums; C2 % 64 == 0 keeps the alignment)
                         slabs=(rawG, raw_product["ZS"], raw_product["splits"]) if rawG is not None else None)
    if rawG is not None:
        rawP = (rawP, rawG, raw_product["ZS"])       # everything the side stream reads from main-stream tensors
    return (g, z2, w, pre, direct, out, side, raw, partial, slab, splits, t, colz, C, C2, ws_t, rawP)


def _bn3_linear_wgrad_finish(state, coef_t, coeffs, coef_late=False):
    """Second half: the split-K reduction that applies the coefficients.  ``coef_late``: the side stream
    was ordered behind the main one BEFORE the coefficients were computed there - wait again."""
    g, z2, w, pre, direct, out, side, raw, partial, slab, splits, t, colz, C, C2, ws_t, rawP = state
    if (coef_late or rawP is not None) and raw is not None:
        K.stream_wait_stream(raw, K._raw_stream(), _SIDE["event"])
    K.splitk_reduce_lin(partial, out, slab, splits, slab, direct, coef_t.data_ptr(), C, t, colz, C2,
                        coeffs.data_ptr(), stream=raw)
    if side is not None:
        # everything the side stream still reads after this function returns - including the saved
        # coefficient block: autograd releases a node's saved tensors when its backward returns, and
        # the allocator would hand their memory to the main stream while the reduction is still queued
        _side_mark_used()
        if _SIDE["join_queued"]:
            _SIDE["keep"].append((g, z2, coef_t, coeffs, pre, rawP))
        else:
            extra = () if rawP is None else (rawP if isinstance(rawP, tuple) else (rawP,))
            for t_ in (g, z2, coef_t, coeffs) + ((pre,) if pre is not None else ()) + extra:
                t_.record_stream(side)
    if direct:
        _notify(w)
        return None
    return out


def _bn3_linear_wgrad(g, z2, w, coef_t, coeffs, prewaited=False, pre=None, coef_late=False, raw_product=None):
    """dW3 by linearity, into ``w.grad`` when that is a live arena view (returns None and notifies),
    else returned.  Runs on the weight-gradient stream like _conv_wgrad: raw stream handle, one
    persistent workspace carved by address (no torch stream switch, no allocations).  ``pre``: the
    Gram part already computed during the forward pass (:func:`_bn3_linear_gram_forward`).
    ``coeffs``: the BatchNorm's saved [4, C] block (its first row, the batch mean, is read here)."""
    return _bn3_linear_wgrad_finish(_bn3_linear_wgrad_start(g, z2, w, prewaited, pre, raw_product), coef_t, coeffs, coef_late)


class _RawOut:
    """Output operand given by raw address (a slice of a workspace) for launchers that ask the
    tensor for ``data_ptr()`` / ``numel()``."""

    __slots__ = ("ptr", "n")

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n

    def data_ptr(self):
        return self.ptr

    def numel(self):
        return self.n


class ConvFn(Function):
    """y = conv2d(x, w) (no bias); optionally also the per-tile BatchNorm partial sums."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, want_stats):
        x = _as_nhwc(x)
        if not _krsc_ok(w):
            raise K.KernelError("conv weight must be channels_last (KRSC) in memory")
        y, stats = _conv_fwd(x, w, stride, padding, want_stats)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding)
        if stats is None:
            stats = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _unused):
        x, w = ctx.saved_tensors
        stride, padding = ctx.cfg
        dy = _as_nhwc(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad(dy, w, x.shape, stride, padding)
        if ctx.needs_input_grad[1]:
            Kc, _, R, S = w.shape
            dw = _wgrad_into(dy, x, _geom(x), torch.empty_like(w), Kc, R, S, stride, padding,
                             padding, dy.shape[2], dy.shape[3])
        return dx, dw, None, None, None


def _stem_dw(dy, xp, geom, w, cin, Ho, Wo):
    """Filter gradient of the stem: split-K GEMM into the padded [K][7][8][4] fp32 image, then one
    kernel that drops the padding, casts and - in direct mode - accumulates straight into ``w.grad``
    (the gradient arena; returns None and notifies the exchange), else returns a new tensor."""
    Kc = w.shape[0]
    dwp = torch.empty((Kc, 7, 8, 4), dtype=torch.float32, device=dy.device)
    _wgrad_into(dy, xp, geom, dwp, Kc, 7, 1, 2, 3, 0, Ho, Wo)
    if _grad_listeners is not None and _DIRECT["on"] and _direct_grad_ok(w) and _krsc_ok(w):
        K.stem_dw_unpack(dwp, w.grad, Kc, cin, True)
        _notify(w)
        return None
    dw = torch.empty_like(w)
    K.stem_dw_unpack(dwp, dw, Kc, cin, False)
    return dw


class StemConvFn(Function):
    """7x7 / stride 2 / pad 3 convolution of a 3-channel image (K5).  The image is repacked to a
    W-padded NHWC4 layout so that each filter row is one contiguous 32-element run; the conv
    then runs on the common implicit-GEMM kernel with R = 7 "taps" of 32 channels."""

    @staticmethod
    def forward(ctx, x, w, want_stats):
        if x.dtype != _BF16:
            x = x.to(_BF16)
        B, C, H, W = x.shape
        Kc = w.shape[0]
        xp = torch.empty((B, H, W + 8, 4), dtype=_BF16, device=x.device)
        K.stem_pack(x, xp)
        wp = torch.empty((Kc, 7, 8, 4), dtype=_BF16, device=x.device)
        K.stem_weight_pack(w, wp)
        Ho = (H + 6 - 7) // 2 + 1
        Wo = (W + 6 - 7) // 2 + 1
        geom = (B, H, W + 1, 32, H * (W + 8) * 4, (W + 8) * 4, 4, xp.numel())
        y = _empty_nhwc(B, Kc, Ho, Wo, x.device)
        stats = None
        if want_stats:
            stats = torch.empty((K.conv_tiles_m(B * Ho * Wo), 2, Kc), dtype=torch.float32,
                                device=x.device)
        K.conv_igemm(xp, geom, wp, y, Kc, 7, 1, 2, 3, 0, 1, Ho, Wo, stats=stats)
        ctx.save_for_backward(xp, w)
        ctx.geom = geom
        ctx.cin = C
        if stats is None:
            stats = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _unused):
        xp, w = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise K.KernelError("the stem convolution does not provide a gradient for its input")
        dy = _as_nhwc(dy)
        Kc = w.shape[0]
        Ho, Wo = dy.shape[2], dy.shape[3]
        return None, _stem_dw(dy, xp, ctx.geom, w, ctx.cin, Ho, Wo), None


class StemFn(Function):
    """The whole ResNet stem in training mode as one node (``HorovodTF/src/resnet_model.py:
    256-266``): 7x7/2 conv (K5) -> BatchNorm -> ReLU -> 3x3/2 max pool, with the BatchNorm apply,
    the ReLU and the pooling in ONE pass over the conv output — the 112x112x64 activation (411 MB
    at batch 256) is never written.  Backward recomputes the pool gradient from the pooled
    gradient and the argmax bytes inside the two BatchNorm-backward passes."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, rmean, rvar, momentum, eps):
        if x.dtype != _BF16:
            x = x.to(_BF16)
        B, C, H, W = x.shape
        Kc = w.shape[0]
        dev = x.device
        xp = torch.empty((B, H, W + 8, 4), dtype=_BF16, device=dev)
        K.stem_pack(x, xp)
        wp = torch.empty((Kc, 7, 8, 4), dtype=_BF16, device=dev)
        K.stem_weight_pack(w, wp)
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        geom = (B, H, W + 1, 32, H * (W + 8) * 4, (W + 8) * 4, 4, xp.numel())
        M = B * Ho * Wo
        T = K.conv_tiles_m(M)
        P = K._fold_slices(T) if T > K.bn_direct_max() else 0
        y = _empty_nhwc(B, Kc, Ho, Wo, dev)
        coeffs = torch.empty((4, Kc), dtype=torch.float32, device=dev)
        tmp = torch.empty(2 * T * Kc + 4 * P * Kc, dtype=torch.float32, device=dev)
        st = tmp.data_ptr()
        K.conv_igemm(xp, geom, wp, y, Kc, 7, 1, 2, 3, 0, 1, Ho, Wo, stats=st)
        cb = coeffs.data_ptr()
        K.bn_fwd_finalize(st, T, Kc, M, 128, gamma, beta, rmean, rvar, momentum, eps,
                          cb, cb + 4 * Kc, cb + 8 * Kc, cb + 12 * Kc,
                          work=st + 8 * T * Kc if P else None)
        Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        out = _empty_nhwc(B, Kc, Hp, Wp, dev)
        idx = torch.empty((B, Hp, Wp, Kc), dtype=torch.uint8, device=dev)
        K.bn_relu_maxpool_fwd(y, cb + 8 * Kc, cb + 12 * Kc, out, idx, B, Ho, Wo, Kc)
        ctx.save_for_backward(xp, w, y, idx, coeffs, gamma, beta)
        ctx.geom = geom
        ctx.cin = C
        return out

    @staticmethod
    def backward(ctx, dout):
        xp, w, y, idx, coeffs, gamma, beta = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise K.KernelError("the stem does not provide a gradient for its input")
        dout = _as_nhwc(dout)
        B, Kc, Ho, Wo = y.shape
        M = B * Ho * Wo
        dev = y.device
        G = min(max(M // 64, 1), _STEM_REDUCE_BLOCKS)
        P = K._fold_slices(G) if G > K.bn_direct_max() else 0
        tmp = torch.empty(3 * Kc + 2 * G * Kc + 2 * P * Kc, dtype=torch.float32, device=dev)
        coef = tmp.data_ptr()
        part = coef + 12 * Kc
        wk = part + 8 * G * Kc if P else None
        cb = coeffs.data_ptr()
        fco = cb + 8 * Kc
        K.stem_pool_bn_bwd(dout, idx, y, fco, None, part, None, B, Ho, Wo, Kc, G, False)
        direct = _DIRECT["on"] and _direct_grad_ok(gamma) and _direct_grad_ok(beta) and \
            gamma.grad.dtype == torch.float32 and beta.grad.dtype == torch.float32
        if direct:
            K.bn_bwd_finalize(part, G, Kc, M, gamma, cb, cb + 4 * Kc, gamma.grad, beta.grad, coef,
                              accumulate=True, work=wk)
            dg = db = None
        else:
            dgb = torch.empty((2, Kc), dtype=torch.float32, device=dev)
            K.bn_bwd_finalize(part, G, Kc, M, gamma, cb, cb + 4 * Kc, dgb[0], dgb[1], coef, work=wk)
            dg, db = dgb[0], dgb[1]
        dy = torch.empty_like(y)
        K.stem_pool_bn_bwd(dout, idx, y, fco, coef, None, dy, B, Ho, Wo, Kc,
                           min(max(M // 64, 1), _STEM_APPLY_BLOCKS), True)
        if direct:
            _notify(gamma)
            _notify(beta)
        dw = _stem_dw(dy, xp, ctx.geom, w, ctx.cin, Ho, Wo)
        return None, dw, dg, db, None, None, None, None


# timing-only probe for the plan in profiles/r1_notes.md ("BatchNorm backward of the wide unit by
# linearity"): DDL_PROBE_SKIP_WIDE_APPLY=1 drops that apply pass, gradients are then garbage
_PROBE_SKIP_WIDE_APPLY = _os.environ.get("DDL_PROBE_SKIP_WIDE_APPLY", "0") == "1"

# blocks of the stem's fused reduce pass (one [2][C] partial row each)
_STEM_REDUCE_BLOCKS = int(_os.environ.get("DDL_STEM_REDUCE_BLOCKS", "4096"))
_STEM_APPLY_BLOCKS = int(_os.environ.get("DDL_STEM_APPLY_BLOCKS", "4096"))   # blocks of the apply pass (2048-8192 measured equal, 16384+ slower)


def stem_supported(weight, stride, padding):
    return _is_stem(weight, stride, padding) and weight.shape[0] % 8 == 0 and \
        bn_supported(weight.shape[0])


def stem(x, weight, gamma, beta, rmean, rvar, momentum, eps):
    return StemFn.apply(x, weight, gamma, beta, rmean, rvar, momentum, eps)


def conv2d_stats(x, weight, stride, padding, want_stats):
    """(y, stats) where stats is the [tiles, 2, Cout] partial-sum tensor (or None)."""
    if _is_stem(weight, stride, padding):
        y, st = StemConvFn.apply(x, weight, want_stats)
    else:
        C = weight.shape[1]
        if C % 32 != 0 or weight.shape[0] % 32 != 0:
            raise K.KernelError("conv2d HIP kernel needs Cin %% 32 == 0 and Cout %% 32 == 0 "
                                "(got Cin=%d, Cout=%d)" % (C, weight.shape[0]))
        y, st = ConvFn.apply(x, weight, stride, padding, want_stats)
    return y, (st if want_stats else None)


def conv_supported(weight, stride, padding):
    if _is_stem(weight, stride, padding):
        return True
    return weight.shape[1] % 32 == 0 and weight.shape[0] % 32 == 0


def conv2d(x, weight, stride, padding):
    return conv2d_stats(x, weight, stride, padding, False)[0]


# ----------------------------------------------------------------------------------------------
# batch norm (+ residual) (+ ReLU)
# ----------------------------------------------------------------------------------------------
def bn_supported(C):
    return C % 8 == 0 and C // 8 <= 256 and 256 % (C // 8) == 0


class BatchNormActFn(Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, residual, stats, rmean, rvar, training, momentum, eps, relu):
        y = _as_nhwc(y)
        if residual is not None:
            residual = _as_nhwc(residual)
        z, coeffs = _bn_fwd(y, gamma, beta, residual, stats, rmean, rvar, training, momentum, eps,
                            relu)
        ctx.relu = bool(relu)
        ctx.training = bool(training)
        ctx.has_res = residual is not None
        ctx.save_for_backward(y, z if relu else None, gamma, coeffs)
        return z

    @staticmethod
    def backward(ctx, dz):
        if not ctx.training:
            raise K.KernelError("BatchNorm backward in eval mode is not supported by the HIP path")
        y, z, gamma, coeffs = ctx.saved_tensors
        dz = _as_nhwc(dz)
        want_res = ctx.has_res and ctx.needs_input_grad[3]
        dy, dg, db, gres = _bn_bwd(dz, y, z, gamma, coeffs, ctx.relu, want_res, has_res=ctx.has_res)
        return dy, dg, db, gres, None, None, None, None, None, None, None


def batchnorm_act(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu,
                  residual, stats=None):
    return BatchNormActFn.apply(x, gamma, beta, residual, stats, running_mean, running_var,
                                training, momentum, eps, relu)


# ----------------------------------------------------------------------------------------------
# fused residual blocks: one autograd node per block, hand-scheduled backward
# ----------------------------------------------------------------------------------------------
_DS_FWD_SIDE = {"on": _os.environ.get("DDL_DS_FWD_SIDE", "1") != "0",
                "max_hw": int(_os.environ.get("DDL_DS_FWD_SIDE_MAX_HW", "1000000"))}


def _ds_forward_stream(x):
    """Raw handle of the stream the projection shortcut of a block runs on in the forward pass, or
    None for the current stream (no second stream in use, graph capture, DDL_DS_FWD_SIDE=0)."""
    if not (_DS_FWD_SIDE["on"] and _SIDE["on"] and _SIDE_LEAN and x.is_cuda) or x.shape[2] * x.shape[3] > _DS_FWD_SIDE["max_hw"]:
        return None
    if torch.cuda.is_current_stream_capturing():
        return None
    return _side_stream(x.device).cuda_stream


class ResidualBlockFn(Function):
    """A whole bottleneck (3 conv-BN pairs) or basic block (2 pairs) plus optional projection
    shortcut as ONE autograd node (``HorovodTF/src/resnet_model.py:112-204``).

    What the hand-written backward buys over autograd's composition of the per-op nodes:
    * the shortcut gradient is added in the epilogue of the first conv's backward-data kernel
      instead of by a separate elementwise pass over the (largest) block-input tensor;
    * weight and BatchNorm gradients are written straight into the flat gradient arena
      (``param.grad`` views) — the wgrad reduction's output IS the all-reduce buffer;
    * no autograd bookkeeping between the ~30 kernels of a block.

    ``units`` = number of conv-BN pairs on the main path (3 or 2); parameters arrive flat:
    for each unit (weight, gamma, beta, running_mean, running_var), then the same five for the
    projection shortcut when ``has_ds``.
    """

    @staticmethod
    def forward(ctx, x, cfg, *tensors):
        units, strides, pads, has_ds, ds_stride, training, momentum, eps = cfg[:8]
        x = _as_nhwc(x)
        n = units + (1 if has_ds else 0)
        groups = [tensors[5 * i: 5 * i + 5] for i in range(n)]
        saved_y, saved_z, saved_c = [], [], []
        shortcut = x
        yd = cd = None
        ds_raw = None
        if has_ds:
            wd, gd, bd, rmd, rvd = groups[units]
            if training:
                # the projection's BatchNorm is applied inside the last unit's residual add: its
                # normalised output is never materialised (one write + one read of the block's
                # widest tensor less)
                ds_raw = _ds_forward_stream(x)
                if ds_raw is None:
                    yd, _, cd = _unit_fwd(x, wd, ds_stride, 0, gd, bd, None, rmd, rvd, momentum,
                                          eps, False, apply=False)
                else:
                    # the projection only meets the main path at the block's final residual add: it
                    # runs on the (idle in forward) weight-gradient stream next to conv1 .. bn2
                    K.stream_wait_stream(ds_raw, K._raw_stream(), _SIDE["event"])
                    ds_keep = []    # scratch of the side-stream launches: alive until the main stream has waited
                    with K.launch_on(ds_raw):
                        yd, _, cd = _unit_fwd(x, wd, ds_stride, 0, gd, bd, None, rmd, rvd, momentum,
                                              eps, False, apply=False, keep=ds_keep)
                shortcut = yd
            else:
                yd, st = _conv_fwd(x, wd, ds_stride, 0, training)
                shortcut, cd = _bn_fwd(yd, gd, bd, None, st, rmd, rvd, training, momentum, eps,
                                       False)
        cur = x
        zbits_out = None
        xf_pending = None
        ctx.lin_pre = ctx.lin_pre_ds = ctx.lin_xs = None
        # cfg[9] (grad_premasked) is only set by residual_chain with grad mode on: backward will receive a
        # pre-masked gradient and the BatchNorm partials, i.e. it will take the linear path
        lin_chain = bool(cfg[9]) and _FUSE_BN_REDUCE["on"]
        if training and has_ds and lin_chain and _FUSE_DS_REDUCE["on"] and _lin_ds_static_ok(groups[units][0], ds_stride, x):
            ctx.lin_xs = x if ds_stride == 1 else K.subsample2(x)
            ctx.lin_pre_ds = _bn3_linear_gram_forward(ctx.lin_xs, groups[units][0])
        for i in range(units):
            w, g, b, rm, rv = groups[i]
            last = i == units - 1
            if training:
                zbits = None
                if last and len(cfg) > 11 and cfg[11] is not None and _RELU_BITS["on"]:
                    Ho = (cur.shape[2] + 2 * pads[i] - w.shape[2]) // strides[i] + 1
                    Wo = (cur.shape[3] + 2 * pads[i] - w.shape[3]) // strides[i] + 1
                    zbits = torch.empty(cur.shape[0] * Ho * Wo * w.shape[0] // 8, dtype=torch.uint8,
                                        device=cur.device)
                    zbits_out = zbits
                if last and ds_raw is not None:   # the residual add reads the projection's output
                    K.stream_wait_stream(K._raw_stream(), ds_raw, _SIDE["event"])
                if xf_pending is not None:
                    # the previous unit stopped after its statistics: this conv applies that BatchNorm + ReLU itself
                    z_prev = torch.empty_like(xf_pending[0]) if xf_pending[2] else None
                    y, z, c = _unit_fwd(xf_pending[0], w, strides[i], pads[i], g, b, shortcut if last else None,
                                        rm, rv, momentum, eps, True,
                                        res_coeffs=cd if (last and has_ds) else None, bits=zbits,
                                        xf=(xf_pending[1], z_prev))
                    saved_z[-1] = z_prev      # (None: backward hands the raw tensor + coefficients to the weight gradient)
                    xf_pending = None
                elif _ROWS_BN_IN and units >= 2 and i == 0 and not last and \
                        _rows_bn_in_ok(cur, w, strides[i], pads[i], groups[1][0], strides[1], pads[1]):
                    y, z, c = _unit_fwd(cur, w, strides[i], pads[i], g, b, None, rm, rv, momentum, eps, True, apply=False)
                    # (2: only where the row-streaming weight gradient will serve the next conv in backward)
                    wn = groups[1][0]
                    full = _ROWS_BN_IN == 2 and wn.requires_grad and \
                        _grad_listeners is not None and _DIRECT["on"] and _direct_grad_ok(wn) and \
                        K.conv_wgrad_rows_ok(wn.shape[0], wn.shape[1], 3, 3, y.shape[2], y.shape[3], 1, y.shape[0])
                    xf_pending = (y, c, not full)
                else:
                    y, z, c = _unit_fwd(cur, w, strides[i], pads[i], g, b, shortcut if last else None,
                                        rm, rv, momentum, eps, True,
                                        res_coeffs=cd if (last and has_ds) else None, bits=zbits)
            else:
                y, st = _conv_fwd(cur, w, strides[i], pads[i], training)
                z, c = _bn_fwd(y, g, b, shortcut if last else None, st, rm, rv, training,
                               momentum, eps, True)
            saved_y.append(y)
            saved_z.append(z)
            saved_c.append(c)
            if training and lin_chain and units == 3 and i == units - 2 and \
                    _lin_main_static_ok(groups[units - 1][0], strides[units - 1], pads[units - 1]):
                # this block's last BatchNorm will go by linearity in backward: its Gram part only
                # needs z (the narrow activation just written) and runs now, on the idle side stream
                ctx.lin_pre = _bn3_linear_gram_forward(z, groups[units - 1][0])
            cur = z if xf_pending is None else y     # (a pending unit hands its raw conv output on)
        if len(cfg) > 11 and cfg[11] is not None:
            cfg[11].y = saved_y[-1]
            cfg[11].y2 = yd if (has_ds and training) else None
            cfg[11].zbits = zbits_out
        ctx.cfg = cfg
        ctx.n_in = 2 + len(tensors)
        ctx.params = groups  # parameters/buffers are leaves: plain references are safe
        keep = [x] + saved_y + saved_z + saved_c
        if has_ds:
            keep += [yd, cd]
        ctx.save_for_backward(*keep)
        return cur

    @staticmethod
    def backward(ctx, dout):
        units, strides, pads, has_ds, ds_stride, training, momentum, eps = ctx.cfg[:8]
        premask_in, grad_premasked = ctx.cfg[8], ctx.cfg[9]
        link_in, link_out = (ctx.cfg[10], ctx.cfg[11]) if len(ctx.cfg) > 11 else (None, None)
        if not training:
            raise K.KernelError("fused residual block: backward in eval mode is not supported")
        saved = ctx.saved_tensors
        x = saved[0]
        ys = saved[1: 1 + units]
        zs = saved[1 + units: 1 + 2 * units]
        cs = saved[1 + 2 * units: 1 + 3 * units]
        groups = ctx.params
        grads = [None] * ctx.n_in
        dout = _as_nhwc(dout)

        def put(unit, dw, dg, db):
            base = 2 + 5 * unit
            grads[base], grads[base + 1], grads[base + 2] = dw, dg, db

        _DEFER["on"] = True
        try:
            return ResidualBlockFn._backward_body(ctx, saved, x, ys, zs, cs, groups, grads, dout, put)
        finally:
            _DEFER["on"] = False
            _flush_notifications()

    @staticmethod
    def _backward_body(ctx, saved, x, ys, zs, cs, groups, grads, dout, put):
        units, strides, pads, has_ds, ds_stride, training, momentum, eps = ctx.cfg[:8]
        premask_in, grad_premasked = ctx.cfg[8], ctx.cfg[9]
        link_in, link_out = (ctx.cfg[10], ctx.cfg[11]) if len(ctx.cfg) > 11 else (None, None)
        dz = dout
        g_short = None
        # BatchNorm-backward partials produced by the previous dgrad's epilogue (for the last
        # unit: by the NEXT block's input-gradient kernel, handed over through the chain link)
        fused = fused_ds = None
        if grad_premasked and link_out is not None:
            fused, link_out.partial = link_out.partial, None
            fused_ds, link_out.partial2 = link_out.partial2, None
        for i in range(units - 1, -1, -1):
            w, gam, bet = groups[i][0], groups[i][1], groups[i][2]
            last = i == units - 1
            if fused is not None or (last and grad_premasked):
                mask = 0            # dz already carries the ReLU mask
            else:
                mask = 1 if last else 2
            inp = zs[i - 1] if i > 0 else x
            inp_fcoef = None
            if i > 0 and inp is None:
                # the previous unit's activated output was never written (DDL_CONV_ROWS_BN_IN=2): the weight
                # gradient gets the raw conv output and that BatchNorm's scale / shift
                inp = ys[i - 1]
                inp_fcoef = cs[i - 1].data_ptr() + 8 * inp.shape[1]
            if last and i > 0 and mask == 0 and fused is not None and _FUSE_BN_REDUCE["on"] and \
                    _bn3_linear_ok(w, strides[i], pads[i], dz, inp):
                # the wide unit by linearity: its dy never exists (see _bn3_linear_dgrad)
                early = _LIN_EARLY_SIDE > 0 and _side_prewait(dz.device)   # the raw product g^T z2 needs no coefficients
                wstate = None
                if early and _LIN_EARLY_SIDE > 1:    # ... and is issued before the chain's kernels
                    wstate = _bn3_linear_wgrad_start(dz, inp, w, prewaited=True, pre=getattr(ctx, "lin_pre", None))
                coef_t, dg, db = _unit_coef(dz, ys[i], gam, bet, cs[i], fused)
                g_short = dz
                # the dependent chain first (the host is what the main stream waits for in the late
                # layers: issuing the side-stream work first left 40-140 us holes in front of the prep kernel)
                pre = early or _side_prewait(dz.device)   # the side stream may start before the chain's next kernels finish
                # (the raw product - and the Gram part, unless the forward pass did it - may ride in the input-gradient kernel)
                lf = {"want_gram": getattr(ctx, "lin_pre", None) is None} if wstate is None else None
                dz, fused = _bn3_linear_dgrad(g_short, inp, w, coef_t, ys[i - 1],
                                              cs[i - 1].data_ptr() + 8 * inp.shape[1], fuse=lf)
                if wstate is None:
                    wstate = _bn3_linear_wgrad_start(g_short, inp, w, prewaited=pre, pre=getattr(ctx, "lin_pre", None),
                                                     raw_product=lf if (lf and "P" in lf) else None)
                dw = _bn3_linear_wgrad_finish(wstate, coef_t, cs[i], coef_late=bool(early))
                put(i, dw, dg, db)
                _flush_notifications()
                continue
            if i > 0 and mask == 0 and fused is not None and _FUSE_BN_REDUCE["on"] and \
                    _rows_bn_bwd_ok(dz, w, strides[i], pads[i], inp, ys[i - 1]):
                # the apply pass of this unit's BatchNorm backward happens inside the two kernels that read its
                # result: only the coefficients are computed here
                coef_t, dg, db = _unit_coef(dz, ys[i], gam, bet, cs[i], fused)
                if last:
                    g_short = dz
                pre = _side_prewait(dz.device)
                g_in = dz
                dz, fused = _conv_dgrad_bn(g_in, w, inp.shape, strides[i], pads[i], ys[i - 1],
                                           cs[i - 1].data_ptr() + 8 * inp.shape[1], dy_y=ys[i], dy_coef=coef_t)
                dw = _conv_wgrad(g_in, inp, w, strides[i], pads[i], prewaited=pre, x_fcoef=inp_fcoef,
                                 dy_y=ys[i], dy_coef=coef_t, keep=(cs[i - 1],))
                put(i, dw, dg, db)
                _flush_notifications()
                continue
            dy, dg, db, g = _unit_bwd(dz, ys[i], zs[i], gam, bet, cs[i], mask, last, partial=fused)
            if last:
                g_short = g
            if i > 0:
                # chain first: the side stream is ordered behind dy now (prewait), the next kernel
                # of the dependent chain is issued, and only then the weight-gradient kernels - in
                # the late layers the host is what the main stream waits for (40-55 us holes in
                # front of every backward-data kernel otherwise, profiles/r2_notes.md)
                pre = _side_prewait(dy.device)
                if _FUSE_BN_REDUCE["on"]:
                    dz, fused = _conv_dgrad_bn(dy, w, inp.shape, strides[i], pads[i], ys[i - 1],
                                               cs[i - 1].data_ptr() + 8 * inp.shape[1])
                else:
                    dz, fused = _conv_dgrad(dy, w, inp.shape, strides[i], pads[i]), None
                dw = _conv_wgrad(dy, inp, w, strides[i], pads[i], prewaited=pre, x_fcoef=inp_fcoef,
                                 keep=(cs[i - 1],) if inp_fcoef is not None else ())
                put(i, dw, dg, db)
                _flush_notifications()  # the chain's next kernel is queued: now talk to the exchange
            else:
                dw = _conv_wgrad(dy, inp, w, strides[i], pads[i])
                put(i, dw, dg, db)
                addend, sub2 = g_short, False
                if has_ds:
                    wd, gd, bd = groups[units][0], groups[units][1], groups[units][2]
                    yd, cd = saved[1 + 3 * units], saved[2 + 3 * units]
                    even = x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                    if fused_ds is not None and ctx.needs_input_grad[0] and _lin_ds_static_ok(wd, ds_stride, x) and \
                            g_short.shape[1] == wd.shape[0]:
                        # the projection shortcut's BatchNorm by linearity as well (yd = x_s Wd^T with
                        # x_s the input at the pixels the strided 1x1 conv visits): dyd never exists
                        xs = getattr(ctx, "lin_xs", None)
                        if xs is None:
                            xs = x if ds_stride == 1 else K.subsample2(x)
                        coef_d, dgd, dbd = _unit_coef(g_short, yd, gd, bd, cd, fused_ds)
                        pre = _side_prewait(g_short.device)
                        lfd = {"want_gram": getattr(ctx, "lin_pre_ds", None) is None}
                        addend, _ = _bn3_linear_dgrad(g_short, xs, wd, coef_d, None, None, fuse=lfd)
                        sub2 = ds_stride == 2
                        dwd = _bn3_linear_wgrad(g_short, xs, wd, coef_d, cd, prewaited=pre,
                                                pre=getattr(ctx, "lin_pre_ds", None), raw_product=lfd if "P" in lfd else None)
                        put(units, dwd, dgd, dbd)
                    else:
                        dyd, dgd, dbd, _ = _unit_bwd(g_short, yd, None, gd, bd, cd, 0, False,
                                                     partial=fused_ds)
                        dwd = _conv_wgrad(dyd, x, wd, ds_stride, 0)
                        put(units, dwd, dgd, dbd)
                        if ctx.needs_input_grad[0]:
                            if ds_stride == 2 and even:
                                addend, sub2 = _conv_dgrad_1x1s2_compact(dyd, wd), True
                            else:
                                addend = _conv_dgrad(dyd, wd, x.shape, ds_stride, 0)
                if ctx.needs_input_grad[0]:
                    bn_y = link_in.y if (premask_in and link_in is not None and
                                         _FUSE_BN_REDUCE["on"]) else None
                    if bn_y is not None and bn_y.shape != x.shape:
                        bn_y = None
                    bn_y2 = link_in.y2 if (bn_y is not None and _FUSE_DS_REDUCE["on"]) else None
                    if bn_y2 is not None and bn_y2.shape != x.shape:
                        bn_y2 = None
                    zbits = link_in.zbits if (premask_in and link_in is not None) else None
                    if zbits is not None and zbits.numel() * 8 != x.numel():
                        zbits = None
                    res = _conv_dgrad(dy, w, x.shape, strides[i], pads[i], addend=addend,
                                      addend_sub2=sub2, mask_src=x if premask_in else None,
                                      bn_y=bn_y, bn_y2=bn_y2, mask_bits=zbits)
                    if bn_y2 is not None:
                        grads[0], link_in.partial, link_in.partial2 = res
                    elif bn_y is not None:
                        grads[0], link_in.partial = res
                    else:
                        grads[0] = res
        return tuple(grads)


def residual_block_supported(convs):
    return all(w.dim() == 4 and w.shape[1] % 32 == 0 and w.shape[0] % 32 == 0 and
               bn_supported(w.shape[0]) and _krsc_ok(w) for w in convs)


_FUSE_BN_REDUCE = {"on": True}
# the projection shortcut's BatchNorm partials ride in the same epilogue (DDL_FUSE_DS_REDUCE=0:
# separate reduce pass over the block's output gradient, the A/B switch)
_FUSE_DS_REDUCE = {"on": _os.environ.get("DDL_FUSE_DS_REDUCE", "1") != "0"}
# the ReLU mask between chained blocks travels as one bit per element (written by the forward
# BatchNorm apply) instead of being re-read from the bf16 activation (DDL_RELU_BITS=0: re-read)
_RELU_BITS = {"on": _os.environ.get("DDL_RELU_BITS", "1") != "0"}


class _ChainLink:
    """Hand-over between two adjacent fused blocks of a chain: the earlier block leaves the
    pre-BatchNorm output of its last unit (``y``) for the later block, whose backward returns
    the BatchNorm-backward partial sums (``partial``) computed in its input-gradient epilogue."""

    __slots__ = ("y", "partial", "y2", "partial2", "zbits")

    def __init__(self):
        self.y = None
        self.partial = None
        self.y2 = None          # the earlier block's projection-shortcut conv output, if it has one
        self.partial2 = None    # ... and the partials of that BatchNorm (same gradient, other input)
        self.zbits = None       # ReLU mask of the earlier block's output, one bit per element


def residual_block(x, units, downsample, training, momentum, eps, premask_in=False,
                   grad_premasked=False, link_in=None, link_out=None):
    """``units``: list of (weight, stride, padding, gamma, beta, running_mean, running_var);
    ``downsample``: None or (weight, stride, gamma, beta, running_mean, running_var).

    ``premask_in`` / ``grad_premasked`` are the contract of :func:`residual_chain` (never set them
    by hand): the block's input is the ReLU output of the previous fused block, so this block
    returns its input gradient already multiplied by that ReLU's mask, and the previous block is
    told that the gradient it receives is pre-masked (its last BatchNorm backward then reads
    neither the activation nor writes a masked copy)."""
    tensors, strides, pads = [], [], []
    for (w, s, p, g, b, rm, rv) in units:
        tensors += [w, g, b, rm, rv]
        strides.append(s)
        pads.append(p)
    ds_stride = 1
    if downsample is not None:
        w, ds_stride, g, b, rm, rv = downsample
        tensors += [w, g, b, rm, rv]
    cfg = (len(units), tuple(strides), tuple(pads), downsample is not None, ds_stride,
           bool(training), float(momentum), float(eps), bool(premask_in), bool(grad_premasked),
           link_in, link_out)
    return ResidualBlockFn.apply(x, cfg, *tensors)


def residual_chain(x, blocks, training, momentum, eps):
    """Run consecutive residual blocks (``blocks`` = list of (units, downsample) as for
    :func:`residual_block`) whose intermediate activations have no other consumer.  Each block
    is still one autograd node; adjacent nodes exchange ReLU-masked gradients (see
    :func:`residual_block`), which removes two full tensor passes per block boundary."""
    n = len(blocks)
    chain = bool(training) and torch.is_grad_enabled()
    links = [_ChainLink() for _ in range(n - 1)] if chain else []
    for i, (units, ds) in enumerate(blocks):
        x = residual_block(x, units, ds, training, momentum, eps,
                           premask_in=chain and i > 0, grad_premasked=chain and i < n - 1,
                           link_in=links[i - 1] if chain and i > 0 else None,
                           link_out=links[i] if chain and i < n - 1 else None)
    return x


def set_direct_grads(on: bool):
    """Let fused blocks write parameter gradients straight into ``param.grad`` arena views."""
    _DIRECT["on"] = bool(on)


# ----------------------------------------------------------------------------------------------
# pooling
# ----------------------------------------------------------------------------------------------
class MaxPool3x3s2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _as_nhwc(x)
        B, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = _empty_nhwc(B, C, Ho, Wo, x.device)
        idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
        K.maxpool_fwd(x, out, idx, B, H, W, C)
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        dy = _as_nhwc(dy)
        dx = _empty_nhwc(B, C, H, W, dy.device)
        K.maxpool_bwd(dy, idx, dx, B, H, W, C)
        return dx


def maxpool3x3s2(x):
    return MaxPool3x3s2Fn.apply(x)


class GlobalAvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _as_nhwc(x)
        B, C, H, W = x.shape
        out = torch.empty((B, C), dtype=_BF16, device=x.device)
        K.avgpool_fwd(x, out, B, H * W, C)
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = dy.contiguous()
        dx = _empty_nhwc(B, C, H, W, dy.device)
        K.avgpool_bwd(dy, dx, B, H * W, C)
        return dx


def global_avgpool(x):
    return GlobalAvgPoolFn.apply(x)


# ----------------------------------------------------------------------------------------------
# dense head
# ----------------------------------------------------------------------------------------------
def linear_supported(weight):
    return weight.shape[1] % 32 == 0 and weight.shape[0] % 8 == 0


class LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        B, C = x.shape
        N = w.shape[0]
        y = torch.empty((B, N), dtype=_BF16, device=x.device)
        geom = (B, 1, 1, C, C, C, C, x.numel())
        K.conv_igemm(x, geom, w, y, N, 1, 1, 1, 0, 0, 1, 1, 1)
        if bias is not None:
            K.bias_add_(y, bias, B, N)
        ctx.save_for_backward(x, w)
        ctx.bias = bias   # a parameter (leaf): a plain reference is safe
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        bias = ctx.bias
        B, C = x.shape
        N = w.shape[0]
        npad = (N + 31) // 32 * 32
        if dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) == npad and dy.dtype == _BF16 and \
                _XENT["padded"] == (dy.data_ptr(), B, npad):
            # the loss hands over its gradient inside a zero-padded [B, npad] buffer: no pad pass
            dyp = torch.as_strided(dy, (B, npad), (npad, 1))
            _XENT["padded"] = None
        elif npad != N:
            dy = dy.contiguous()
            dyp = torch.zeros((B, npad), dtype=_BF16, device=dy.device)
            dyp[:, :N] = dy
        else:
            dyp = dy.contiguous()
        dx = dw = db = None
        gy = (B, 1, 1, npad, npad, npad, npad, dyp.numel())
        if ctx.needs_input_grad[0]:
            wt, wrows = _dgrad_w(w)
            dx = torch.empty((B, C), dtype=_BF16, device=dy.device)
            K.conv_igemm(dyp, gy, wt, dx, C, 1, 1, 1, 0, 0, 1, 1, 1, wt_rows=wrows)
        direct = _grad_listeners is not None and _DIRECT["on"]
        if ctx.needs_input_grad[1]:
            gx = (B, 1, 1, C, C, C, C, x.numel())
            if direct and _direct_grad_ok(w) and w.is_contiguous():
                _wgrad_into(dyp, x, gx, w.grad, npad, 1, 1, 1, 0, 0, 1, 1, n_valid=N * C, accumulate=True)
                _notify(w)
            else:
                dw = _wgrad_into(dyp, x, gx, torch.empty_like(w), npad, 1, 1, 1, 0, 0, 1, 1,
                                 n_valid=N * C)
        if bias is not None and ctx.needs_input_grad[2]:
            if direct and _direct_grad_ok(bias) and bias.grad.dtype == torch.float32:
                K.colsum(dyp, bias.grad, B, N, npad, accumulate=True)
                _notify(bias)
            else:
                db = torch.empty(N, dtype=torch.float32, device=dy.device)
                K.colsum(dyp, db, B, N, npad)
        return dx, dw, db


def linear(x, weight, bias):
    return LinearFn.apply(x, weight, bias)


# ----------------------------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------------------------
_XENT = {"ticket": None, "padded": None}


def _xent_ticket(device):
    t = _XENT["ticket"]
    if t is None or t.device != device:
        t = _XENT["ticket"] = torch.zeros(4, dtype=torch.int32, device=device)
    return t


class SoftmaxCrossEntropyFn(Function):
    """``F.cross_entropy`` (mean) of the reference (``imagenet_pytorch_horovod.py:340``) in two
    launches per step: forward = row losses and their mean (the last block reduces, fixed order);
    backward = the gradient, recomputed from the logits with the upstream scalar read on the
    device, written into a zero-padded buffer the dense head consumes without a pad pass."""

    @staticmethod
    def forward(ctx, logits, target):
        logits = logits.contiguous()
        if logits.dtype != _BF16:
            logits = logits.to(_BF16)
        target = target.contiguous()
        B, N = logits.shape
        out = torch.empty(B + 1, dtype=torch.float32, device=logits.device)   # row losses, then their mean
        K.softmax_xent(logits, target, out, None, B, N, N, N, 1.0 / B,
                       loss_mean=out.data_ptr() + 4 * B, ticket=_xent_ticket(logits.device))
        ctx.save_for_backward(logits, target)
        return out[B]

    @staticmethod
    def backward(ctx, go):
        logits, target = ctx.saved_tensors
        B, N = logits.shape
        npad = (N + 31) // 32 * 32
        if go.dtype != torch.float32 or not go.is_cuda:
            go = go.to(device=logits.device, dtype=torch.float32)
        buf = torch.empty((B, npad), dtype=_BF16, device=logits.device)
        K.softmax_xent(logits, target, None, buf, B, N, N, npad, 1.0 / B, go=go)
        # columns N..npad of the rows are zeros: LinearFn.backward recognises the buffer by address and
        # shape and reads it whole (valid for this backward pass only)
        _XENT["padded"] = (buf.data_ptr(), B, npad)
        return (buf[:, :N] if npad != N else buf), None


def softmax_cross_entropy(logits, target):
    return SoftmaxCrossEntropyFn.apply(logits, target)
